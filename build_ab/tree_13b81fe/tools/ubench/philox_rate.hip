// Microbenchmark: issue cost of one Philox4x32-10 call in the sweep kernel's form (lane-varying index in
// counter word 1, call index in word 3, everything else wave-uniform), and of its ingredients, at 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../pyisingmontecarlo_amd/csrc/philox.hpp"
using namespace isingmc;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
constexpr int ITER = 256;

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint2 key, uint32_t t)
{
    const uint32_t Q = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (int it = 0; it < ITER; it++) {
        if (MODE == 0) { // 8 calls sharing (t, Q), like the 7 planes + tie call of one quad
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const uint4 r = philox4x32_10(make_uint4(t + it, Q, 0x4C415453u, p), key);
                acc ^= r.x ^ r.y ^ r.z ^ r.w;
            }
        }
        if (MODE == 1) { // 8 calls, nothing shared (call index in a multiplied word)
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const uint4 r = philox4x32_10(make_uint4(t + it, Q, p, 0x4C415453u), key);
                acc ^= r.x ^ r.y ^ r.z ^ r.w;
            }
        }
        if (MODE == 2) { // 128 dependent-free mads (8 independent chains of 16)
            uint32_t a[8];
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = Q + i + it;
#pragma unroll
            for (int j = 0; j < 16; j++)
#pragma unroll
                for (int i = 0; i < 8; i++) { const uint64_t p = uint64_t(a[i]) * 0xD2511F53u; a[i] = uint32_t(p >> 32) ^ uint32_t(p); }
#pragma unroll
            for (int i = 0; i < 8; i++) acc ^= a[i];
        }
        if (MODE == 3) { // 128 x (bitop3 with an SGPR operand), 8 independent chains
            uint32_t a[8];
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = Q + i + it;
#pragma unroll
            for (int j = 0; j < 16; j++)
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = __builtin_amdgcn_bitop3_b32(a[i], a[(i + 1) & 7], key.x + j, 0x96);
#pragma unroll
            for (int i = 0; i < 8; i++) acc ^= a[i];
        }
    }
    out[Q] = acc;
}

template <int MODE>
int run(const char *name, double units_per_iter, uint32_t *d)
{
    const int blocks = 256 * 8 * 4; // 4 rounds of 8 workgroups (4 waves each) per CU = 8 waves/SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, make_uint2(123, 456), 7u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, make_uint2(123, 456), 7u);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (waves_per_simd * ITER * units_per_iter);
    printf("%-44s %8.3f ms  %7.2f cycles per unit per wave (SIMD issue time @2.4 GHz)\n", name, ms, cyc);
    return 0;
}

int main()
{
    uint32_t *d; CK(hipMalloc(&d, 256 * 8 * 4 * 256 * 4));
    run<0>("philox call, shared rounds 2-3 (unit = call)", 8, d);
    run<1>("philox call, nothing shared (unit = call)", 8, d);
    run<2>("v_mad_u64_u32 + v_xor (unit = pair)", 128, d);
    run<3>("v_bitop3 (v, v, s) (unit = instr)", 128, d);
    return 0;
}
