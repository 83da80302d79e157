// Translation unit of the one-degree replica-packed kernels (packed_types.hpp).
#include "packed_uni_kernels.hpp"

#include <type_traits>

namespace isingmc {

template <int D>
static void launch_degree(bool uniform_beta, bool mixed_signs, dim3 grid, hipStream_t stream, uint32_t *state, const PkGraphDev &G,
                          const PkUniHeaders &H, uint32_t class_begin, uint32_t class_end, uint64_t t, const uint2 *group_keys,
                          const uint32_t *tabs, uint32_t tab_stride, bool table)
{
    const auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, dim3(256), 0, stream, state, G, H, class_begin, class_end, t, group_keys, tabs, tab_stride);
    };
#ifdef ISINGMC_PKU_TABLE_ALWAYS // A/B builds: the instantiation with table reads for every class
    table = true;
#endif
    const auto pick = [&](auto ub, auto pmj) {
        if (table) launch(pk_sweep_uni_kernel<D, decltype(ub)::value, decltype(pmj)::value, true>);
        else launch(pk_sweep_uni_kernel<D, decltype(ub)::value, decltype(pmj)::value, false>);
    };
    if (uniform_beta) {
        if (mixed_signs) pick(std::true_type{}, std::true_type{});
        else pick(std::true_type{}, std::false_type{});
    } else {
        if (mixed_signs) pick(std::false_type{}, std::true_type{});
        else pick(std::false_type{}, std::false_type{});
    }
}

hipError_t pk_uni_launch_sweep(int degree, bool uniform_beta, bool mixed_signs, uint32_t n_groups, hipStream_t stream, uint32_t *state,
                               const PkGraphDev &G, const PkUniHeaders &H, uint32_t class_begin, uint32_t class_end, uint64_t t,
                               const uint2 *group_keys, const uint32_t *tabs, uint32_t tab_stride, bool needs_table)
{
    const dim3 grid((class_end - class_begin + 1023) / 1024, n_groups); // 256 threads x 4 positions per workgroup
    switch (degree) {
    case 3: launch_degree<3>(uniform_beta, mixed_signs, grid, stream, state, G, H, class_begin, class_end, t, group_keys, tabs, tab_stride, needs_table); break;
    case 4: launch_degree<4>(uniform_beta, mixed_signs, grid, stream, state, G, H, class_begin, class_end, t, group_keys, tabs, tab_stride, needs_table); break;
    case 5: launch_degree<5>(uniform_beta, mixed_signs, grid, stream, state, G, H, class_begin, class_end, t, group_keys, tabs, tab_stride, needs_table); break;
    case 6: launch_degree<6>(uniform_beta, mixed_signs, grid, stream, state, G, H, class_begin, class_end, t, group_keys, tabs, tab_stride, needs_table); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace isingmc
