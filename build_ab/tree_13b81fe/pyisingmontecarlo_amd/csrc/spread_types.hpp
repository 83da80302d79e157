// Launcher of the lane-parallel resident lattice kernel (spread_kernels.hpp; own translation unit, see there).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace isingmc {

struct LatGeom;
struct LatThr;

// arguments as lat_resident_kernel; lanes_per_quad = 8, 4 or 2; threads >= lanes_per_quad * nquads, lds_bytes = the two planes + 128 bytes per quad
hipError_t spread_launch(bool vec, bool pmj, int lanes_per_quad, unsigned blocks, unsigned threads, size_t lds_bytes, hipStream_t stream, uint32_t *state,
                         const LatGeom &g, uint64_t t0, uint32_t timesteps, const uint2 *keys, const LatThr *thr_steps, uint32_t thr_stride,
                         const LatThr *thr_replica, const uint32_t *jneg, uint32_t jneg_uniform, unsigned long long *steps_out,
                         uint32_t n_replicas);
// workgroups of that instantiation one compute unit holds at once (the runtime's occupancy calculation); 0 when the query fails
int spread_blocks_per_cu(bool vec, bool pmj, int lanes_per_quad, unsigned threads, size_t lds_bytes);

} // namespace isingmc
