// Shared declarations of the replica-packed REAL-COUPLING path (real_kernels.hpp): any f64 couplings and
// any site biases -- the edge list of lattice.rs:46-50 with set_individual_bias / set_global_bias (lattice.rs:104-131)
// materialised as lattice.rs:186-189 does -- on graphs of degree <= 15.  Own translation unit (real_kernels.hip), like the
// strip and multi-class kernels, so that the streaming lattice kernels keep their register allocation.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace isingmc {

constexpr int RJ_MAX_DEG = 15;
constexpr uint32_t DOM_RJ_SWEEP = 0x524A5357u; // "RJSW"
constexpr int RJ_LOG_INTERVALS = 2048;         // intervals of the log2(1 + m) table (Q24 values, linear interpolation)

// ELL view of the quantised graph, slot-major (a wave's 64 positions read 256 contiguous bytes per slot).
// Couplings and biases are integers in units of 2^k (k chosen once per graph, host_logic.cpp rj_quantise).
struct RjGraphDev {
    const uint32_t *nbr; // [slots][n_pos] neighbour position; the own position in the unused slots of a site
    const int32_t *jq;   // [slots][n_pos] quantised coupling; 0 in unused slots
    const int32_t *hq;   // [n_pos] quantised bias (0 on padding)
    const uint2 *logtab; // [RJ_LOG_INTERVALS] {LT[i], LT[i+1] - LT[i]}
    uint32_t n_pos;      // multiple of 256
    uint32_t slots;      // 4, 7, 11 or 15: the smallest of them that holds the largest degree
};

// acceptance scale of one inverse temperature: accept iff max(X >> shift, 0) <= (Lambda_q(u) * mant) >> 32
struct RjBeta {
    uint32_t shift, mant;
};

// One colour class of one timestep.  Positions [class_begin, real_end) are the real sites of the class (the padding behind
// them is left alone); grid.y = replica groups.  betas: beta_stride == 0: one RjBeta for every replica of the launch;
// beta_stride == 32: betas[32 g + b] for replica bit b of group g.
// q_lo, q_hi: the Philox calls (groups of four replica bits) to decide, 0 .. 8 = all of them (see rj_sweep_kernel PARTIAL)
hipError_t rj_launch_sweep(dim3 grid, hipStream_t stream, uint32_t *state, const RjGraphDev &G, uint32_t class_begin, uint32_t real_end,
                           uint64_t t, const uint2 *group_keys, const RjBeta *betas, uint32_t beta_stride, uint32_t q_lo = 0, uint32_t q_hi = 8);

// out[2 slot] += -2 x (energy of replica slot in units of 2^k, bias terms included) as an int64 in two's complement,
// out[2 slot + 1] += up spins; slot = 32 g + b.  class0_end: 0, or -- on a graph of exactly two colour classes -- the end
// of class 0: the bonds are then counted from the class-0 positions alone.  scan_end: positions [0, scan_end) are visited
// (n_pos; class0_end when the graph is two-coloured, has no biases and the up spins are not wanted).  Padding carries the
// PAD marker in `site`.
// count_up: also the up spins (32 more registers per thread); energy-only callers pass false and leave out[2 slot + 1] alone.
hipError_t rj_launch_measure(dim3 grid, hipStream_t stream, const uint32_t *state, const RjGraphDev &G, const uint32_t *site,
                             uint32_t class0_end, uint32_t scan_end, bool count_up, unsigned long long *out);
// on-stream tempering: out[r] = energy of slot first_slot + r from the counters a measurement left in meas
hipError_t rj_launch_energy_from_counts(hipStream_t stream, unsigned long long *meas, uint32_t first_slot, uint32_t n, int k,
                                        double self_energy, double *out);
// threads per workgroup of the kernels for `slots` (256, or 128 for 11 / 15 slots: their per-thread tables are 48 / 64 words)
uint32_t rj_threads(uint32_t slots);
// workgroups of that instantiation one CU holds (the launch is sized so that all of them are resident at once)
int rj_measure_blocks_per_cu(uint32_t slots, bool bipartite, bool count_up);

} // namespace isingmc
