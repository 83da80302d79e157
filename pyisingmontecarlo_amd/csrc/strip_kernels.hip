// Translation unit of the persistent strip kernel (see strip_types.hpp for why it is apart from isingmc.hip).
#include "strip_kernels.hpp"

namespace isingmc {

hipError_t strip_launch(bool pmj, int nw, unsigned blocks, size_t lds_bytes, hipStream_t stream, uint32_t *state, const LatGeom &g,
                        const StripArgs &a, uint64_t t0, uint32_t timesteps, const uint2 *keys, const LatThr *thr_steps,
                        uint32_t thr_stride, const LatThr *thr_replica, const uint32_t *jneg, uint32_t jneg_uniform,
                        unsigned long long *halo, unsigned long long *steps_out, const StripFinal &fin, uint32_t n_replicas,
                        uint32_t *err)
{
    const auto launch = [&](auto kernel, unsigned threads) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds_bytes, stream, state, g, a, t0, timesteps, keys, thr_steps, thr_stride,
                           thr_replica, jneg, jneg_uniform, halo, steps_out, fin, n_replicas, err);
    };
    if (nw == 1) { if (pmj) launch(lat_strip_kernel<true, 1>, 64u); else launch(lat_strip_kernel<false, 1>, 64u); }
    else { if (pmj) launch(lat_strip_kernel<true, 4>, 256u); else launch(lat_strip_kernel<false, 4>, 256u); }
    return hipGetLastError();
}

} // namespace isingmc
