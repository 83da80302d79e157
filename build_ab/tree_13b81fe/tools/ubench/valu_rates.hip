// Microbenchmark: issue cost of the integer instructions Philox4x32 is made of, on gfx950.
// Each kernel runs ITER x 16 independent ops per thread on 8 waves/SIMD; reports cycles per
// wave-instruction per SIMD = (elapsed cycles x SIMDs) / (waves x instructions per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
constexpr int ITER = 2048;

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 8 + i;
    uint64_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) { a[i] = a[i] ^ (a[i] >> 3) ^ seed; }                       // 2 cheap ops (xor3/bitop + shift)
            if (OP == 1) { uint64_t p = uint64_t(a[i]) * 0xD2511F53u; a[i] = uint32_t(p >> 32) ^ uint32_t(p); } // mad_u64 + xor
            if (OP == 2) { a[i] = __umulhi(a[i], 0xD2511F53u) ^ seed; }              // mul_hi + xor
            if (OP == 3) { a[i] = a[i] * 0xD2511F53u ^ seed; }                       // mul_lo + xor
            if (OP == 4) { a[i] = __umul24(a[i], 0x511F53u) ^ seed; }                // mul_u32_u24 + xor
            if (OP == 5) { a[i] = a[i] ^ seed; a[i] += it; }                         // 2 cheap ops
            if (OP == 6) { acc[i] += uint64_t(a[i]) * 0xD2511F53u; a[i] ^= uint32_t(acc[i] >> 32); } // mad_u64 w/ 64-bit add + xor
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i] ^ uint32_t(acc[i]);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int OP>
int run(const char *name, int ops_per_iter, uint32_t *d)
{
    const int blocks = 256 * 8; // 8 blocks of 4 waves per CU = 8 waves/SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double waves = blocks * 4.0, simds = 1024.0;
    double inst_per_wave = double(ITER) * 8 * ops_per_iter;
    double ns_per_inst_simd = ms * 1e6 * simds / (waves * inst_per_wave);
    printf("%-28s %8.3f ms  %6.3f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz) [counting %d instr/elem]\n",
           name, ms, ns_per_inst_simd, ns_per_inst_simd * 2.4, ops_per_iter);
    return 0;
}

int main()
{
    uint32_t *d; CK(hipMalloc(&d, 256 * 8 * 256 * 4));
    run<0>("xor+shift+xor (3 cheap)", 3, d);
    run<5>("xor+add (2 cheap)", 2, d);
    run<1>("mad_u64_u32 + xor", 2, d);
    run<2>("mul_hi_u32 + xor", 2, d);
    run<3>("mul_lo_u32 + xor", 2, d);
    run<4>("mul_u32_u24 + xor", 2, d);
    run<6>("mad_u64_u32(acc) + xor", 2, d);
    return 0;
}
