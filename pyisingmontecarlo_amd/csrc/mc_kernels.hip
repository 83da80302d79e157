// Translation unit of the multi-class checkerboard kernels (mc_types.hpp).
#include "mc_kernels.hpp"

namespace isingmc {

hipError_t mc_launch_sweep(int mode, bool pmj, dim3 grid, hipStream_t stream, uint32_t *state, const LatGeom &g, uint32_t colour,
                           uint64_t t, const uint2 *keys, const LatThrMC &thr_uniform, const LatThrMC *thr_replica,
                           const uint32_t *jneg, uint32_t jneg_uniform, McOpen open)
{
    const auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, dim3(256), 0, stream, state, g, colour, t, keys, thr_uniform, thr_replica, jneg, jneg_uniform, open);
    };
    if (mode == MC_FIELD) { if (pmj) launch(lat_mc_sweep_kernel<MC_FIELD, true>); else launch(lat_mc_sweep_kernel<MC_FIELD, false>); }
    else if (mode == MC_ANISO) { if (pmj) launch(lat_mc_sweep_kernel<MC_ANISO, true>); else launch(lat_mc_sweep_kernel<MC_ANISO, false>); }
    else { if (pmj) launch(lat_mc_sweep_kernel<MC_OPEN, true>); else launch(lat_mc_sweep_kernel<MC_OPEN, false>); }
    return hipGetLastError();
}

hipError_t mc_launch_measure_aniso(bool pmj, dim3 grid, hipStream_t stream, const uint32_t *state, const LatGeom &g, const uint32_t *jneg,
                                   uint32_t jneg_uniform, unsigned long long *out, size_t out_stride)
{
    if (pmj) hipLaunchKernelGGL(lat_mc_measure_aniso_kernel<true>, grid, dim3(256), 0, stream, state, g, jneg, jneg_uniform, out, out_stride);
    else hipLaunchKernelGGL(lat_mc_measure_aniso_kernel<false>, grid, dim3(256), 0, stream, state, g, jneg, jneg_uniform, out, out_stride);
    return hipGetLastError();
}

hipError_t mc_launch_measure_open(bool pmj, dim3 grid, hipStream_t stream, const uint32_t *state, const LatGeom &g, const uint32_t *jneg,
                                  uint32_t jneg_uniform, McOpen open, unsigned long long *out, size_t out_stride)
{
    if (pmj) hipLaunchKernelGGL(lat_mc_measure_open_kernel<true>, grid, dim3(256), 0, stream, state, g, jneg, jneg_uniform, open, out, out_stride);
    else hipLaunchKernelGGL(lat_mc_measure_open_kernel<false>, grid, dim3(256), 0, stream, state, g, jneg, jneg_uniform, open, out, out_stride);
    return hipGetLastError();
}

} // namespace isingmc
