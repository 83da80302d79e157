// Translation unit of the replica-packed real-coupling kernels (see real_types.hpp for why it is apart from isingmc.hip).
#include "real_kernels.hpp"

#include <type_traits>

namespace isingmc {

template <typename F>
static void rj_with_slots(uint32_t slots, F &&f)
{
    if (slots == 4) f(std::integral_constant<int, 4>{});
    else if (slots == 7) f(std::integral_constant<int, 7>{});
    else if (slots == 11) f(std::integral_constant<int, 11>{});
    else if (slots == 15) f(std::integral_constant<int, 15>{});
    else if (slots == 23) f(std::integral_constant<int, 23>{});
    else f(std::integral_constant<int, 31>{});
}

uint32_t rj_threads(uint32_t slots)
{
    uint32_t n = 256;
    rj_with_slots(slots, [&](auto s_c) { n = uint32_t(RjShape<decltype(s_c)::value>::THREADS); });
    return n;
}

hipError_t rj_launch_sweep(dim3 grid, hipStream_t stream, uint32_t *state, const RjGraphDev &G, uint32_t class_begin, uint32_t real_end,
                           uint64_t t, const uint2 *group_keys, const RjBeta *betas, uint32_t beta_stride, uint32_t q_lo, uint32_t q_hi)
{
    rj_with_slots(G.slots, [&](auto s_c) {
        constexpr int S = decltype(s_c)::value;
        const auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, grid, dim3(RjShape<S>::THREADS), 0, stream, state, G, class_begin, real_end, t, group_keys, betas,
                               q_lo, q_hi);
        };
        const bool whole = q_lo == 0 && q_hi >= 8; // every replica bit of the groups of this launch is decided
        const bool heavy = G.dshift != nullptr;
        // (partly owned groups of graphs WITHOUT heavy sites have their own instantiations: carrying the heavy-site shift cost the
        //  few-replica runs 9 % -- 2048^2 Gaussian x 8: 4.98 -> 4.54e11 -- when round 4 first folded them into the heavy ones)
        const auto pick = [&](auto ub_c) {
            constexpr bool UB = decltype(ub_c)::value;
            if (!whole) { if (heavy) launch(rj_sweep_kernel<S, UB, true, true>); else launch(rj_sweep_kernel<S, UB, true, false>); }
            else if (heavy) launch(rj_sweep_kernel<S, UB, false, true>);
            else launch(rj_sweep_kernel<S, UB, false, false>);
        };
        if (beta_stride == 0) pick(std::true_type{});
        else pick(std::false_type{});
    });
    return hipGetLastError();
}

template <typename F>
static void rj_pick_measure(uint32_t slots, bool bip, bool up, F &&f)
{
    rj_with_slots(slots, [&](auto s_c) {
        constexpr int S = decltype(s_c)::value;
        if (bip) { if (up) f(rj_measure_kernel<S, true, true>, RjShape<S>::THREADS); else f(rj_measure_kernel<S, true, false>, RjShape<S>::THREADS); }
        else { if (up) f(rj_measure_kernel<S, false, true>, RjShape<S>::THREADS); else f(rj_measure_kernel<S, false, false>, RjShape<S>::THREADS); }
    });
}

hipError_t rj_launch_measure(dim3 grid, hipStream_t stream, const uint32_t *state, const RjGraphDev &G, const uint32_t *site,
                             uint32_t class0_end, uint32_t scan_end, bool count_up, unsigned long long *out)
{
    rj_pick_measure(G.slots, class0_end != 0, count_up, [&](auto kernel, int threads) {
        hipLaunchKernelGGL(kernel, grid, dim3(threads), 0, stream, state, G, site, class0_end, scan_end, out);
    });
    return hipGetLastError();
}

hipError_t rj_launch_energy_from_counts(hipStream_t stream, unsigned long long *meas, uint32_t first_slot, uint32_t n, int k_energy,
                                        double self_energy, double *out)
{
    hipLaunchKernelGGL(rj_energy_from_counts_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, meas, first_slot, n, k_energy, self_energy, out);
    return hipGetLastError();
}

int rj_measure_blocks_per_cu(uint32_t slots, bool bipartite, bool count_up)
{
    int n = 0;
    rj_pick_measure(slots, bipartite, count_up, [&](auto kernel, int threads) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, 0) != hipSuccess) n = 0;
    });
    (void)hipGetLastError();
    return n;
}

} // namespace isingmc
