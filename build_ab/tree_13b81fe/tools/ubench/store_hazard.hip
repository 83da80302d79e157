// Microbenchmark / reproducer (round 3): a buffer_store_dwordx4 whose data registers are overwritten by the next vector-ALU
// instruction.  LLVM's hazard recogniser (GCNHazardRecognizer::createsVALUHazard) inserts wait states behind a > 64-bit MUBUF
// store only when its soffset operand is an IMMEDIATE; with a REGISTER soffset it inserts none.  On a loaded MI355X the
// register form then stores the overwritten value for part of the wave (lat_sweep_measure_kernel: bit counts instead of spins).
//   mode 0: register soffset, v_bcnt on a data register right behind the store            (the pattern that failed)
//   mode 1: immediate soffset, same source                                                 (the compiler inserts the wait states)
//   mode 2..: register soffset, s_nop (mode - 2) between the store and the overwrite      (how many wait states are enough)
// Every thread stores `iters` 16-byte records of a known pattern; a second kernel counts records that differ from it.
//   hipcc --offload-arch=gfx950 -O3 -o store_hazard store_hazard.hip && ./store_hazard
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x | 0x80000001u;
}

// LOADS: four 16-byte loads and one 4-byte load from a second buffer (all zeros) in front of every store, as the sweep kernels
// issue them -- the vector memory pipeline then holds loads of this and of the other waves when the store arrives
template <int MODE, bool LOADS>
__global__ __launch_bounds__(256) void store_kernel(uint32_t *buf, const uint32_t soff, const uint32_t iters, const uint32_t bytes,
                                                    uint32_t *sink, const uint32_t *zeros)
{
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, n = gridDim.x * 256;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(buf, 0, int(bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t zsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(zeros), 0, int(bytes), 0x00020000);
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t rec = it * n + gid;
        u32x4 d = {mix(4 * rec), mix(4 * rec + 1), mix(4 * rec + 2), mix(4 * rec + 3)};
        if constexpr (LOADS) {
            const uint32_t o = rec * 16;
            const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(zsrc, o, 0, 0);
            const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(zsrc, (o + 4096) % bytes, 0, 0);
            const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(zsrc, (o + bytes - 4096) % bytes, 0, 0);
            const u32x4 e = __builtin_amdgcn_raw_buffer_load_b128(zsrc, (o + bytes / 2) % bytes, 0, 0);
            const uint32_t f = __builtin_amdgcn_raw_buffer_load_b32(zsrc, (o + 16) % bytes, 0, 0);
            d.x ^= a.x ^ b.y ^ c.z ^ e.w ^ f; d.y ^= a.y ^ b.z ^ c.w ^ e.x; d.z ^= a.z ^ b.w ^ c.x ^ e.y; d.w ^= a.w ^ b.x ^ c.y ^ e.z;
        }
        if constexpr (MODE == 1) {
            __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, rec * 16, 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, rec * 16 - soff, soff, 0);     // soff: a kernel argument, i.e. an SGPR
            if constexpr (MODE == 3) asm volatile("s_nop 0");
            if constexpr (MODE == 4) asm volatile("s_nop 1");
            if constexpr (MODE == 5) asm volatile("s_nop 3");
            if constexpr (MODE == 6) asm volatile("s_nop 7");
        }
        asm volatile("v_bcnt_u32_b32 %0, %0, 0" : "+v"(d.z));       // overwrites the third data register
        acc += d.z;
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;                             // keeps the counts alive
}

__global__ void check_kernel(const uint32_t *buf, const uint32_t records, unsigned long long *bad, unsigned long long *bad_word2_is_count)
{
    const uint32_t rec = blockIdx.x * 256 + threadIdx.x;
    if (rec >= records) return;
    const uint4 v = reinterpret_cast<const uint4 *>(buf)[rec];
    const uint32_t w0 = mix(4 * rec), w1 = mix(4 * rec + 1), w2 = mix(4 * rec + 2), w3 = mix(4 * rec + 3);
    if (v.x != w0 || v.y != w1 || v.z != w2 || v.w != w3) {
        atomicAdd(bad, 1ull);
        if (v.x == w0 && v.y == w1 && v.w == w3 && v.z == uint32_t(__popc(w2))) atomicAdd(bad_word2_is_count, 1ull);
    }
}

static uint32_t *g_zeros = nullptr;
static bool g_loads = false;

template <int MODE>
static int run(const char *what, uint32_t *buf, uint32_t blocks, uint32_t iters, unsigned long long *d_bad, uint32_t *d_sink)
{
    const uint32_t records = blocks * 256 * iters, bytes = records * 16;
    CK(hipMemset(buf, 0, bytes));
    CK(hipMemset(d_bad, 0, 16));
    if (g_loads) hipLaunchKernelGGL((store_kernel<MODE, true>), dim3(blocks), dim3(256), 0, 0, buf, 0u, iters, bytes, d_sink, g_zeros);
    else hipLaunchKernelGGL((store_kernel<MODE, false>), dim3(blocks), dim3(256), 0, 0, buf, 0u, iters, bytes, d_sink, g_zeros);
    hipLaunchKernelGGL(check_kernel, dim3((records + 255) / 256), dim3(256), 0, 0, buf, records, d_bad, d_bad + 1);
    unsigned long long h[2];
    CK(hipMemcpy(h, d_bad, 16, hipMemcpyDeviceToHost));
    printf("%-62s blocks %6u x %u records/thread: %10llu of %10u records wrong (%llu of them: third word = its own bit count)\n", what, blocks, iters,
           h[0], records, h[1]);
    return 0;
}

int main()
{
    uint32_t *buf, *d_sink;
    unsigned long long *d_bad;
    const uint32_t max_blocks = 65536, iters = 4;
    CK(hipMalloc(&buf, size_t(max_blocks) * 256 * iters * 16));
    CK(hipMalloc(&d_bad, 16));
    CK(hipMalloc(&d_sink, 4));
    CK(hipMalloc(&g_zeros, size_t(max_blocks) * 256 * iters * 16));
    CK(hipMemset(g_zeros, 0, size_t(max_blocks) * 256 * iters * 16));
    for (int loads = 0; loads < 2; loads++)
    for (uint32_t blocks : {256u, 2048u, 65536u}) {
        g_loads = loads != 0;
        printf("---- %s\n", g_loads ? "five loads in front of every store" : "stores only");
        if (run<0>("register soffset, overwrite right behind the store", buf, blocks, iters, d_bad, d_sink)) return 1;
        if (run<1>("immediate soffset (compiler's wait states)", buf, blocks, iters, d_bad, d_sink)) return 1;
        if (run<3>("register soffset + s_nop 0 (1 wait state)", buf, blocks, iters, d_bad, d_sink)) return 1;
        if (run<4>("register soffset + s_nop 1 (2 wait states)", buf, blocks, iters, d_bad, d_sink)) return 1;
        if (run<5>("register soffset + s_nop 3 (4 wait states)", buf, blocks, iters, d_bad, d_sink)) return 1;
        if (run<6>("register soffset + s_nop 7 (8 wait states)", buf, blocks, iters, d_bad, d_sink)) return 1;
    }
    return 0;
}
