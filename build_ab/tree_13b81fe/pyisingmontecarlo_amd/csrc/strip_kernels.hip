// Translation unit of the persistent strip kernel (see strip_types.hpp for why it is apart from isingmc.hip).
#include "strip_kernels.hpp"

namespace isingmc {

hipError_t strip_launch(bool pmj, int nw, unsigned blocks, size_t lds_bytes, hipStream_t stream, uint32_t *state, const LatGeom &g,
                        const StripArgs &a, uint64_t t0, uint32_t timesteps, const uint2 *keys, const LatThr *thr_steps,
                        uint32_t thr_stride, const LatThr *thr_replica, const uint32_t *jneg, uint32_t jneg_uniform,
                        unsigned long long *halo, unsigned long long *steps_out, const StripFinal &fin, const StripLadder &lad,
                        uint32_t n_replicas, uint32_t *err)
{
    const auto launch = [&](auto kernel, unsigned threads) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, stream, state, g, a, t0, timesteps, keys, thr_steps, thr_stride,
                           thr_replica, jneg, jneg_uniform, halo, steps_out, fin, lad, n_replicas, err);
    };
    const bool ladder = lad.ladder != nullptr;
    const auto pick = [&](auto pmj_c, auto nw_c) {
        constexpr bool P = decltype(pmj_c)::value;
        constexpr int N = decltype(nw_c)::value;
        if (ladder) launch(lat_strip_kernel<P, N, true>, 64u * N);
        else launch(lat_strip_kernel<P, N, false>, 64u * N);
    };
    if (nw == 1) { if (pmj) pick(std::true_type{}, std::integral_constant<int, 1>{}); else pick(std::false_type{}, std::integral_constant<int, 1>{}); }
    else { if (pmj) pick(std::true_type{}, std::integral_constant<int, 4>{}); else pick(std::false_type{}, std::integral_constant<int, 4>{}); }
    return hipGetLastError();
}

// workgroups of this instantiation that one CU holds at once, by the runtime's own occupancy calculation (registers, the
// launch's dynamic LDS, wave slots); 0 when the query fails
int strip_blocks_per_cu(bool pmj, int nw, bool ladder, size_t lds_bytes)
{
    int n = 0;
    const auto ask = [&](auto kernel, int threads) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, lds_bytes) != hipSuccess) n = 0;
    };
    const auto pick = [&](auto pmj_c, auto nw_c) {
        constexpr bool P = decltype(pmj_c)::value;
        constexpr int N = decltype(nw_c)::value;
        if (ladder) ask(lat_strip_kernel<P, N, true>, 64 * N);
        else ask(lat_strip_kernel<P, N, false>, 64 * N);
    };
    if (nw == 1) { if (pmj) pick(std::true_type{}, std::integral_constant<int, 1>{}); else pick(std::false_type{}, std::integral_constant<int, 1>{}); }
    else { if (pmj) pick(std::true_type{}, std::integral_constant<int, 4>{}); else pick(std::false_type{}, std::integral_constant<int, 4>{}); }
    (void)hipGetLastError();
    return n;
}

} // namespace isingmc
