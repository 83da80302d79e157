// Host-visible types and launchers of the multi-class checkerboard kernels (mc_kernels.hpp): recognised lattices with
// a uniform field (Lattice.set_global_bias, lattice.rs:129-131; ClassicIsing(longitudinal), classicising.rs:69) or with
// open boundaries.  A translation unit of their own, like the strip kernel (strip_types.hpp says why).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace isingmc {

struct LatGeom;

enum : int { MC_NONE = 0, MC_FIELD = 1, MC_OPEN = 2, MC_ANISO = 3, MC_FIELD_OPEN = 4 };
constexpr int MC_MAX_CLASSES = 9;

// Acceptance classes of a spin.  MC_FIELD (periodic, |h| <= 2|J|): class 2 (k - 2) + s for k = 2, 3, 4 satisfied bonds
// and spin bit s -- flipping costs dE = 2|J|(2k - 4) + 2 h (2s - 1); k < 2 always flips.  MC_OPEN (no field, bonds
// across the open boundary absent): class m - 1 for m = satisfied - unsatisfied EXISTING bonds = 1 .. 4, dE = 2|J| m.
// MC_ANISO (periodic, no field, |Jx| != |Jy|): classes by (kx, ky) = satisfied horizontal / vertical bonds:
// 0 (2,2), 1 (2,1), 2 (1,2), 3 (2,0), 4 (0,2); dE = 2 (|Jx| (2 kx - 2) + |Jy| (2 ky - 2)); every other pair has dE <= 0.
// MC_FIELD_OPEN (open boundaries AND a field): with sigma = spin x sign of the site's field, classes 2 (m - 1) + (sigma > 0)
// for m = satisfied - unsatisfied existing bonds = 1 .. 4, and class 8 for m = 0, sigma > 0; dE = 2|J| m + 2|h| sigma.
// Fields of one size and both signs (h_i = +-h, the bimodal random-field model; Lattice.set_individual_bias,
// lattice.rs:104-127): a sign plane per colour (bit set where h_i < 0, each colour's compact layout) turns the spin bit
// into sigma; MC_FIELD and MC_FIELD_OPEN take it (`fneg`, NULL without), then with |h| in the thresholds.
// Per class: the top N_PLANES bits and the low 32 bits of T = floor(exp(-beta dE) 2^THR_BITS); bit c of `costly` is
// clear where the class flips outright (dE <= 0 or T = 2^THR_BITS).
struct LatThrMC {
    uint32_t hi[MC_MAX_CLASSES], lo[MC_MAX_CLASSES];
    uint32_t costly;
};

struct McOpen {
    uint32_t open_x, open_y;
    uint32_t fneg_uniform; // MC_FIELD_OPEN without a sign plane: ~0 when the uniform field is negative
};

// fneg: the field-sign planes [2][wpp] (colour-major) or NULL
hipError_t mc_launch_sweep(int mode, bool pmj, dim3 grid, hipStream_t stream, uint32_t *state, const LatGeom &g, uint32_t colour,
                           uint64_t t, const uint2 *keys, const LatThrMC &thr_uniform, const LatThrMC *thr_replica,
                           const uint32_t *jneg, uint32_t jneg_uniform, McOpen open, const uint32_t *fneg);
// small lattices: `timesteps` whole timesteps of replicas [0, n_replicas) in one launch, the planes in LDS (lds_bytes = both
// planes); thr_steps[k * thr_stride] = the thresholds of timestep t0 + k unless thr_replica (per replica) is given
hipError_t mc_launch_resident(int mode, bool pmj, unsigned n_replicas, unsigned threads, size_t lds_bytes, hipStream_t stream,
                              uint32_t *state, const LatGeom &g, uint64_t t0, uint32_t timesteps, const uint2 *keys,
                              const LatThrMC *thr_steps, uint32_t thr_stride, const LatThrMC *thr_replica, const uint32_t *jneg,
                              uint32_t jneg_uniform, McOpen open, const uint32_t *fneg, unsigned long long *steps_out,
                              uint32_t steps_replicas);
// (steps_out: optional, the measure kernels' two counters after every timestep, [step][steps_replicas][2])
// MC_ANISO: out[r * stride] += satisfied horizontal bonds | satisfied vertical bonds << 32, out[r * stride + 1] += up spins
hipError_t mc_launch_measure_aniso(bool pmj, dim3 grid, hipStream_t stream, const uint32_t *state, const LatGeom &g, const uint32_t *jneg,
                                   uint32_t jneg_uniform, unsigned long long *out, size_t out_stride);
// satisfied EXISTING bonds and up spins per replica: out[r * stride] += sat, out[r * stride + 1] += up; with field-sign
// planes (fneg != NULL) out[r * stride] += sat | (spins along their site's field) << 32
hipError_t mc_launch_measure_open(bool pmj, dim3 grid, hipStream_t stream, const uint32_t *state, const LatGeom &g, const uint32_t *jneg,
                                  uint32_t jneg_uniform, McOpen open, const uint32_t *fneg, unsigned long long *out, size_t out_stride);

} // namespace isingmc
