// Translation unit of the lane-parallel resident lattice kernel (see spread_kernels.hpp for why it is apart from isingmc.hip).
#include "spread_kernels.hpp"

namespace isingmc {

template <int LPQ, typename F>
static void spread_pick_lpq(bool vec, bool pmj, F &&f)
{
    if (vec) { if (pmj) f(lat_resident_spread_kernel<true, true, LPQ>); else f(lat_resident_spread_kernel<true, false, LPQ>); }
    else { if (pmj) f(lat_resident_spread_kernel<false, true, LPQ>); else f(lat_resident_spread_kernel<false, false, LPQ>); }
}

template <typename F>
static void spread_pick(bool vec, bool pmj, int lanes_per_quad, F &&f)
{
    if (lanes_per_quad == 8) spread_pick_lpq<8>(vec, pmj, f);
    else if (lanes_per_quad == 4) spread_pick_lpq<4>(vec, pmj, f);
    else spread_pick_lpq<2>(vec, pmj, f);
}

hipError_t spread_launch(bool vec, bool pmj, int lanes_per_quad, unsigned blocks, unsigned threads, size_t lds_bytes, hipStream_t stream, uint32_t *state,
                         const LatGeom &g, uint64_t t0, uint32_t timesteps, const uint2 *keys, const LatThr *thr_steps, uint32_t thr_stride,
                         const LatThr *thr_replica, const uint32_t *jneg, uint32_t jneg_uniform, unsigned long long *steps_out,
                         uint32_t n_replicas)
{
    spread_pick(vec, pmj, lanes_per_quad, [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, stream, state, g, t0, timesteps, keys, thr_steps, thr_stride,
                           thr_replica, jneg, jneg_uniform, steps_out, n_replicas);
    });
    return hipGetLastError();
}

int spread_blocks_per_cu(bool vec, bool pmj, int lanes_per_quad, unsigned threads, size_t lds_bytes)
{
    int n = 0;
    spread_pick(vec, pmj, lanes_per_quad, [&](auto kernel) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, int(threads), lds_bytes) != hipSuccess) n = 0;
    });
    (void)hipGetLastError();
    return n;
}

} // namespace isingmc
