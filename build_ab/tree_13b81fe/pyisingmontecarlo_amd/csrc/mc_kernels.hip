// Translation unit of the multi-class checkerboard kernels (mc_types.hpp).
#include "mc_kernels.hpp"

namespace isingmc {

template <int MODE>
static void launch_mode(bool pmj, bool uni, dim3 grid, hipStream_t stream, uint32_t *state, const LatGeom &g, uint32_t colour, uint64_t t,
                        const uint2 *keys, const LatThrMC &thr_uniform, const LatThrMC *thr_replica, const uint32_t *jneg,
                        uint32_t jneg_uniform, McOpen open, const uint32_t *fneg)
{
    const auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, dim3(256), 0, stream, state, g, colour, t, keys, thr_uniform, thr_replica, jneg, jneg_uniform, open, fneg);
    };
    if constexpr (MODE == MC_FIELD || MODE == MC_FIELD_OPEN) {
        if (fneg) {
            if (uni) { if (pmj) launch(lat_mc_sweep_kernel<MODE, true, true, true>); else launch(lat_mc_sweep_kernel<MODE, false, true, true>); }
            else { if (pmj) launch(lat_mc_sweep_kernel<MODE, true, false, true>); else launch(lat_mc_sweep_kernel<MODE, false, false, true>); }
            return;
        }
    }
    if (uni) { if (pmj) launch(lat_mc_sweep_kernel<MODE, true, true, false>); else launch(lat_mc_sweep_kernel<MODE, false, true, false>); }
    else { if (pmj) launch(lat_mc_sweep_kernel<MODE, true, false, false>); else launch(lat_mc_sweep_kernel<MODE, false, false, false>); }
}

hipError_t mc_launch_sweep(int mode, bool pmj, dim3 grid, hipStream_t stream, uint32_t *state, const LatGeom &g, uint32_t colour,
                           uint64_t t, const uint2 *keys, const LatThrMC &thr_uniform, const LatThrMC *thr_replica,
                           const uint32_t *jneg, uint32_t jneg_uniform, McOpen open, const uint32_t *fneg)
{
    const bool uni = g.cols_log2 >= 0; // the 2^k mapping of the streaming kernels applies (build_lattice)
#define MC_LAUNCH(M) launch_mode<M>(pmj, uni, grid, stream, state, g, colour, t, keys, thr_uniform, thr_replica, jneg, jneg_uniform, open, fneg)
    if (mode == MC_FIELD) MC_LAUNCH(MC_FIELD);
    else if (mode == MC_ANISO) MC_LAUNCH(MC_ANISO);
    else if (mode == MC_FIELD_OPEN) MC_LAUNCH(MC_FIELD_OPEN);
    else MC_LAUNCH(MC_OPEN);
#undef MC_LAUNCH
    return hipGetLastError();
}

template <int MODE>
static void launch_resident_mode(bool pmj, unsigned n_replicas, unsigned threads, size_t lds_bytes, hipStream_t stream, uint32_t *state,
                                 const LatGeom &g, uint64_t t0, uint32_t timesteps, const uint2 *keys, const LatThrMC *thr_steps,
                                 uint32_t thr_stride, const LatThrMC *thr_replica, const uint32_t *jneg, uint32_t jneg_uniform, McOpen open,
                                 const uint32_t *fneg, unsigned long long *steps_out, uint32_t steps_replicas)
{
    const auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(n_replicas), dim3(threads), lds_bytes, stream, state, g, t0, timesteps, keys, thr_steps, thr_stride,
                           thr_replica, jneg, jneg_uniform, open, fneg, steps_out, steps_replicas);
    };
    // lds_bytes beyond the two planes: room for the spread variant's random words (128 bytes per quad)
    const bool spread = lds_bytes > size_t(2) * g.wpp * sizeof(uint32_t);
    if constexpr (MODE == MC_FIELD || MODE == MC_FIELD_OPEN) {
        if (fneg) {
            if (spread) { if (pmj) launch(lat_mc_resident_kernel<MODE, true, true, true>); else launch(lat_mc_resident_kernel<MODE, false, true, true>); }
            else { if (pmj) launch(lat_mc_resident_kernel<MODE, true, true, false>); else launch(lat_mc_resident_kernel<MODE, false, true, false>); }
            return;
        }
    }
    if (spread) { if (pmj) launch(lat_mc_resident_kernel<MODE, true, false, true>); else launch(lat_mc_resident_kernel<MODE, false, false, true>); }
    else { if (pmj) launch(lat_mc_resident_kernel<MODE, true, false, false>); else launch(lat_mc_resident_kernel<MODE, false, false, false>); }
}

hipError_t mc_launch_resident(int mode, bool pmj, unsigned n_replicas, unsigned threads, size_t lds_bytes, hipStream_t stream,
                              uint32_t *state, const LatGeom &g, uint64_t t0, uint32_t timesteps, const uint2 *keys,
                              const LatThrMC *thr_steps, uint32_t thr_stride, const LatThrMC *thr_replica, const uint32_t *jneg,
                              uint32_t jneg_uniform, McOpen open, const uint32_t *fneg, unsigned long long *steps_out,
                              uint32_t steps_replicas)
{
#define MC_RES(M) launch_resident_mode<M>(pmj, n_replicas, threads, lds_bytes, stream, state, g, t0, timesteps, keys, thr_steps, thr_stride, thr_replica, jneg, jneg_uniform, open, fneg, steps_out, steps_replicas)
    if (mode == MC_FIELD) MC_RES(MC_FIELD);
    else if (mode == MC_ANISO) MC_RES(MC_ANISO);
    else if (mode == MC_FIELD_OPEN) MC_RES(MC_FIELD_OPEN);
    else MC_RES(MC_OPEN);
#undef MC_RES
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
                                  uint32_t jneg_uniform, McOpen open, const uint32_t *fneg, unsigned long long *out, size_t out_stride)
{
    if (pmj) hipLaunchKernelGGL(lat_mc_measure_open_kernel<true>, grid, dim3(256), 0, stream, state, g, jneg, jneg_uniform, open, fneg, out, out_stride);
    else hipLaunchKernelGGL(lat_mc_measure_open_kernel<false>, grid, dim3(256), 0, stream, state, g, jneg, jneg_uniform, open, fneg, out, out_stride);
    return hipGetLastError();
}

} // namespace isingmc
