// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths the kernels here use (MI355X_MICROARCH.md, HBM section:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  One pass over a 1 GiB
// buffer (beyond the 256 MiB Infinity Cache), every byte read exactly once:
//   read_b32   4 bytes per lane per load (a wave reads 256 contiguous bytes; four loads 16 KiB apart per thread, as the packed kernels)
//   read_b128  16 bytes per lane per load (a wave reads 1 KiB: the lattice kernels' quads)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_calib tools/ubench/fetch_calib.hip
// run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- tools/ubench/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ __launch_bounds__(256) void read_b32(const uint32_t *__restrict__ src, uint32_t *__restrict__ out, const uint32_t n_words)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(src), 0, int(0x7FFFFFFF), 0x00020000);
    // thread t of block b reads words 1024 b + t + 256 q, q = 0..3
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x;
    uint32_t acc = 0;
    if (base + 768 < n_words) {
#pragma unroll
        for (int q = 0; q < 4; q++) acc ^= __builtin_amdgcn_raw_buffer_load_b32(r, 4 * (base + 256 * q), 0, 0);
    }
    if (acc == 0x12345678u) out[0] = acc; // (never true for the fill pattern: keeps the loads alive)
}

__global__ __launch_bounds__(256) void read_b128(const uint4 *__restrict__ src, uint32_t *__restrict__ out, const uint32_t n_vec)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t acc = 0;
    if (i < n_vec) {
        const uint4 v = src[i];
        acc = v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    const size_t bytes = size_t(1) << 30; // two kernels over the FIRST 1 GiB window each of its own 1 GiB buffer: byte offsets stay below 2^31
    uint32_t *a = nullptr, *b = nullptr, *out = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(a, 0x5A, bytes);
    (void)hipMemset(b, 0xA5, bytes);
    (void)hipDeviceSynchronize();
    const uint32_t n_words = uint32_t(bytes / 4);
    hipLaunchKernelGGL(read_b32, dim3(n_words / 1024), dim3(256), 0, 0, a, out, n_words);
    hipLaunchKernelGGL(read_b128, dim3(n_words / 4 / 256), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(b), out, n_words / 4);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    std::printf("read_b32 and read_b128 each read %zu bytes once\n", bytes);
    return 0;
}
