// Translation unit of the one-degree replica-packed kernels (packed_types.hpp).
#include "packed_uni_kernels.hpp"

namespace isingmc {

template <int D>
static void launch_degree(bool uniform_beta, bool mixed_signs, dim3 grid, hipStream_t stream, uint32_t *state, const PkGraphDev &G,
                          const PkUniHeaders &H, uint32_t class_begin, uint32_t class_end, uint64_t t, const uint2 *group_keys,
                          const uint32_t *tabs, uint32_t tab_stride)
{
    const auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, dim3(256), 0, stream, state, G, H, class_begin, class_end, t, group_keys, tabs, tab_stride);
    };
    if (uniform_beta) {
        if (mixed_signs) launch(pk_sweep_uni_kernel<D, true, true>);
        else launch(pk_sweep_uni_kernel<D, true, false>);
    } else {
        if (mixed_signs) launch(pk_sweep_uni_kernel<D, false, true>);
        else launch(pk_sweep_uni_kernel<D, false, false>);
    }
}

hipError_t pk_uni_launch_sweep(int degree, bool uniform_beta, bool mixed_signs, dim3 grid, hipStream_t stream, uint32_t *state,
                               const PkGraphDev &G, const PkUniHeaders &H, uint32_t class_begin, uint32_t class_end, uint64_t t,
                               const uint2 *group_keys, const uint32_t *tabs, uint32_t tab_stride)
{
    switch (degree) {
    case 3: launch_degree<3>(uniform_beta, mixed_signs, grid, stream, state, G, H, class_begin, class_end, t, group_keys, tabs, tab_stride); break;
    case 4: launch_degree<4>(uniform_beta, mixed_signs, grid, stream, state, G, H, class_begin, class_end, t, group_keys, tabs, tab_stride); break;
    case 5: launch_degree<5>(uniform_beta, mixed_signs, grid, stream, state, G, H, class_begin, class_end, t, group_keys, tabs, tab_stride); break;
    case 6: launch_degree<6>(uniform_beta, mixed_signs, grid, stream, state, G, H, class_begin, class_end, t, group_keys, tabs, tab_stride); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace isingmc
