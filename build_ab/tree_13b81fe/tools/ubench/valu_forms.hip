// Microbenchmark: SIMD issue time of individual VALU instruction FORMS on gfx950 (inline asm, 8 independent
// chains per thread, 8 waves per SIMD).  cycles = elapsed * 2.4 GHz / (waves per SIMD * groups per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
constexpr int ITER = 512;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define XVV(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define XSV(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(s1));
#define B3VVV(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
#define B3VVS(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b[i]), "s"(s1));
#define AOVVV(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
#define MADVS(i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, 0" : "=v"(w[i]) : "v"(a[i]), "s"(s2) : "s20", "s21");
#define OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
#define ADDVV(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "s"(s2));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(s2));
#define ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b[i]));
#define B3ON_B(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(b[i]) : "v"(b[(i + 1) & 7]), "s"(s1));
#define XON_B(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(b[i]) : "v"(b[(i + 1) & 7]));
#define MADB3(i) MADVS(i) B3ON_B(i)
#define MADX(i) MADVS(i) XON_B(i)
#define MADB3B3(i) MADVS(i) B3ON_B(i) B3VVS(i)
#define MADVV(i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, 0" : "=v"(w[i]) : "v"(a[i]), "v"(b[i]) : "s20", "s21");
#define MULHIV(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define MULLOV(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define XLIT(i) asm volatile("v_xor_b32 %0, 0x9E3779B9, %0" : "+v"(a[i]));
#define ADDLIT(i) asm volatile("v_add_u32 %0, 0x9E3779B9, %0" : "+v"(a[i]));
#define CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
#define SHR(i) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[i]));
#define LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b[i]));
#define BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(a[i]));
#define MOVS(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(s1));
#define MOVV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define NOT(i) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]));
#define FFBL(i) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[i]));
#define BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define XAD(i) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
#define LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b[i]));
#define PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
#define B3VVI(i) asm volatile("v_bitop3_b32 %0, %0, %1, -1 bitop3:0x96" : "+v"(a[i]) : "v"(b[i]));
#define ANDVV(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define CMPV(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
#define XS1(i) XVV(i) asm volatile("s_add_i32 s22, s22, %0" : : "s"(s1) : "s22", "scc");
#define XS2(i) XVV(i) asm volatile("s_add_i32 s22, s22, %0\n s_xor_b32 s23, s23, %0" : : "s"(s1) : "s22", "s23", "scc");
#define MADS1(i) MADVS(i) asm volatile("s_add_i32 s22, s22, %0" : : "s"(s1) : "s22", "scc");
#define MADS2(i) MADVS(i) asm volatile("s_mul_i32 s22, s22, %0\n s_mul_hi_u32 s23, s23, %0" : : "s"(s1) : "s22", "s23", "scc");
#define BRV(i) XVV(i) asm volatile("s_cmp_eq_u32 %0, 77\n s_cbranch_scc1 1f\n s_nop 0\n1:" : : "s"(s1) : "scc");
#define PKADD(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t sv, uint32_t sw)
{
    uint32_t a[8], b[8];
    uint64_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 8 + i; b[i] = a[i] * 3 + sv; w[i] = a[i]; }
    const uint32_t s1 = __builtin_amdgcn_readfirstlane(sv), s2 = __builtin_amdgcn_readfirstlane(sw);
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
            if (MODE == 0) { REP8(XVV) }
            if (MODE == 1) { REP8(XSV) }
            if (MODE == 2) { REP8(B3VVV) }
            if (MODE == 3) { REP8(B3VVS) }
            if (MODE == 4) { REP8(AOVVV) }
            if (MODE == 5) { REP8(MADVS) }
            if (MODE == 6) { REP8(OR3) }
            if (MODE == 7) { REP8(ADDVV) }
            if (MODE == 8) { REP8(MULHI) }
            if (MODE == 9) { REP8(MULLO) }
            if (MODE == 10) { REP8(ALIGN) }
            if (MODE == 11) { REP8(MADB3) }
            if (MODE == 12) { REP8(MADX) }
            if (MODE == 13) { REP8(MADB3B3) }
            if (MODE == 14) { REP8(MADVV) }
            if (MODE == 15) { REP8(MULHIV) }
            if (MODE == 16) { REP8(MULLOV) }
            if (MODE == 17) { REP8(XLIT) }
            if (MODE == 18) { REP8(ADDLIT) }
            if (MODE == 19) { REP8(CNDM) }
            if (MODE == 20) { REP8(SHR) }
            if (MODE == 21) { REP8(LSHLOR) }
            if (MODE == 22) { REP8(BFE) }
            if (MODE == 23) { REP8(MOVS) }
            if (MODE == 24) { REP8(MOVV) }
            if (MODE == 25) { REP8(NOT) }
            if (MODE == 26) { REP8(FFBL) }
            if (MODE == 27) { REP8(BCNT) }
            if (MODE == 28) { REP8(XAD) }
            if (MODE == 29) { REP8(ADD3) }
            if (MODE == 30) { REP8(LSHLADD) }
            if (MODE == 31) { REP8(PERM) }
            if (MODE == 32) { REP8(B3VVI) }
            if (MODE == 33) { REP8(ANDVV) }
            if (MODE == 34) { REP8(CMPV) }
            if (MODE == 35) { REP8(PKADD) }
            if (MODE == 36) { REP8(XS1) }
            if (MODE == 37) { REP8(XS2) }
            if (MODE == 38) { REP8(MADS1) }
            if (MODE == 39) { REP8(MADS2) }
            if (MODE == 40) { REP8(BRV) }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i] ^ b[i] ^ uint32_t(w[i]) ^ uint32_t(w[i] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
int run(const char *name, int per_rep, uint32_t *d)
{
    const int blocks = 256 * 8 * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u, 0xD2511F53u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u, 0xD2511F53u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (waves_per_simd * ITER * 4.0 * 8.0);
    printf("%-44s %8.3f ms  %6.2f cycles per group of %d instruction(s)\n", name, ms, cyc, per_rep);
    fflush(stdout);
    return 0;
}

int main()
{
    uint32_t *d; CK(hipMalloc(&d, 256 * 8 * 4 * 256 * 4));
    run<0>("v_xor_b32 v,v,v", 1, d);
    run<1>("v_xor_b32 v,s,v", 1, d);
    run<2>("v_bitop3_b32 v,v,v,v", 1, d);
    run<3>("v_bitop3_b32 v,v,v,s", 1, d);
    run<4>("v_and_or_b32 v,v,v,v", 1, d);
    run<5>("v_mad_u64_u32 v,s", 1, d);
    run<6>("v_or3_b32 v,v,v,v", 1, d);
    run<7>("v_add_u32 v,v,v", 1, d);
    run<8>("v_mul_hi_u32 v,v,s", 1, d);
    run<9>("v_mul_lo_u32 v,v,s", 1, d);
    run<10>("v_alignbit_b32 v,v,v,1", 1, d);
    run<11>("v_mad_u64_u32 + v_bitop3 (v,v,v,s)", 2, d);
    run<12>("v_mad_u64_u32 + v_xor (v,v,v)", 2, d);
    run<13>("v_mad_u64_u32 + 2 v_bitop3", 3, d);
    run<14>("v_mad_u64_u32 v,v", 1, d);
    run<15>("v_mul_hi_u32 v,v,v", 1, d);
    run<16>("v_mul_lo_u32 v,v,v", 1, d);
    run<17>("v_xor_b32 v,literal,v", 1, d);
    run<18>("v_add_u32 v,literal,v", 1, d);
    run<19>("v_cndmask_b32 v,v,v,vcc", 1, d);
    run<20>("v_lshrrev_b32 v,3,v", 1, d);
    run<21>("v_lshl_or_b32 v,v,3,v", 1, d);
    run<22>("v_bfe_u32 v,v,3,7", 1, d);
    run<23>("v_mov_b32 v,s", 1, d);
    run<24>("v_mov_b32 v,v", 1, d);
    run<25>("v_not_b32", 1, d);
    run<26>("v_ffbl_b32", 1, d);
    run<27>("v_bcnt_u32_b32", 1, d);
    run<28>("v_xad_u32", 1, d);
    run<29>("v_add3_u32", 1, d);
    run<30>("v_lshl_add_u32", 1, d);
    run<31>("v_perm_b32", 1, d);
    run<32>("v_bitop3_b32 v,v,v,-1", 1, d);
    run<33>("v_and_b32 v,v,v", 1, d);
    run<34>("v_cmp_lt_u32 vcc,v,v", 1, d);
    run<35>("v_pk_add_u16 v,v,v", 1, d);
    run<36>("v_xor (v,v,v) + 1 s_add", 2, d);
    run<37>("v_xor (v,v,v) + s_add + s_xor", 3, d);
    run<38>("v_mad_u64_u32 + 1 s_add", 2, d);
    run<39>("v_mad_u64_u32 + s_mul_i32 + s_mul_hi_u32", 3, d);
    run<40>("v_xor + s_cmp + s_cbranch (not taken) + s_nop", 4, d);
    return 0;
}
