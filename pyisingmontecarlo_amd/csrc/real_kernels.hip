// Translation unit of the replica-packed real-coupling kernels (see real_types.hpp for why it is apart from isingmc.hip).
#include "real_kernels.hpp"

namespace isingmc {

hipError_t rj_launch_sweep(dim3 grid, hipStream_t stream, uint32_t *state, const RjGraphDev &G, uint32_t class_begin, uint32_t real_end,
                           uint64_t t, const uint2 *group_keys, const RjBeta *betas, uint32_t beta_stride)
{
    const auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, dim3(RJ_THREADS), 0, stream, state, G, class_begin, real_end, t, group_keys, betas);
    };
    if (G.slots == 4) { if (beta_stride == 0) launch(rj_sweep_kernel<4, true>); else launch(rj_sweep_kernel<4, false>); }
    else { if (beta_stride == 0) launch(rj_sweep_kernel<7, true>); else launch(rj_sweep_kernel<7, false>); }
    return hipGetLastError();
}

hipError_t rj_launch_measure(dim3 grid, hipStream_t stream, const uint32_t *state, const RjGraphDev &G, const uint32_t *site,
                             uint32_t class0_end, unsigned long long *out)
{
    const auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(RJ_THREADS), 0, stream, state, G, site, class0_end, out); };
    const bool bip = class0_end != 0;
    if (G.slots == 4) { if (bip) launch(rj_measure_kernel<4, true>); else launch(rj_measure_kernel<4, false>); }
    else { if (bip) launch(rj_measure_kernel<7, true>); else launch(rj_measure_kernel<7, false>); }
    return hipGetLastError();
}

} // namespace isingmc
