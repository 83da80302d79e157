// Host-visible types and the launcher of the persistent strip kernel (strip_kernels.hpp).  The kernel lives in a
// translation unit of its own (strip_kernels.hip): instantiated next to the streaming kernels in isingmc.hip it
// changed THEIR register allocation (lat_sweep_loop_kernel went from 64 to 66 VGPRs = one wave per SIMD less).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace isingmc {

struct LatGeom;
struct LatThr;

constexpr uint32_t STRIP_MAX_WAVES_PER_CU = 16;           // policy bound (<= 4 blocks of 256 threads per CU: above it the per-colour launches win);
                                                          // the residency bound proper comes from strip_blocks_per_cu()
constexpr unsigned long long STRIP_TIMEOUT_TICKS = 200000000ull; // 2 s of the 100 MHz counter
constexpr uint32_t STRIP_ERR_TIMEOUT = 1u;
constexpr int STRIP_ARRIVAL_SHIFT = 48;                   // final counter word: satisfied bonds | arrived strips << 48

typedef unsigned long long __attribute__((address_space(1))) * strip_gu64;
typedef uint32_t __attribute__((address_space(1))) * strip_gu32;

struct StripArgs {
    uint32_t S;         // rows per strip; S * (wpr / 4) == 64 * NW
    uint32_t n_strips;  // H / S >= 2
    uint32_t qpr_log2;  // log2(wpr / 4)
    uint32_t epoch;     // tags of this launch are epoch + 1 ... epoch + 2 * timesteps (never 0, never reused)
    uint32_t xcd_remap; // n_replicas % 8 == 0: the strips of a replica share blockIdx % 8 (one XCD; speed only)
};

// what the last timestep's measurement becomes (tempering rounds): the strip that arrives last converts the replica's
// satisfied-bond total to its energy E = |J| (n_bonds - 2 sat) and leaves the counter zeroed for the next round
struct StripFinal {
    unsigned long long *counts; // [replica]: satisfied bonds | arrived strips << 48, zero between launches; nullptr: off
    double *energy_out;         // [replica]
    double jabs;
    long long n_bonds;
};

// Parallel-tempering exchange rounds INSIDE the launch (single GPU): after every swap_every-th timestep except the last of
// the launch, the replicas at neighbouring rungs exchange temperatures exactly as pt_swap_kernel / isingmc_host_pt_swap_round
// decide it (same Philox counters (rung, round), same det_exp) -- but pair by pair, with no kernel boundary: the strip of a
// replica that arrives last posts the replica's satisfied-bond total into a mailbox indexed by the replica's RUNG
// (an 8-byte {tag = round + 1, count} granule, as the halo rows), every strip of the two partners polls both mailboxes,
// takes the same decision and relabels itself.  ladder == nullptr: off.
struct StripLadder {
    const double *ladder;                  // beta per rung
    const unsigned long long *ladder_thr;  // {T3, T4} per rung
    const uint32_t *perm_in;               // rung -> slot at launch (not written during the launch)
    uint32_t *perm_out;                    // rung -> slot after the launch's rounds (a different buffer)
    unsigned long long *mail;              // [4][n_rungs] granules, zero before the first use, tags grow with the round
    unsigned long long *round_counts;      // [2][n_replicas]: satisfied bonds | arrived strips << 48 of the round's parity, zero between rounds
    unsigned long long *counters;          // [0] = exchange rounds done (set to round0 + rounds of this launch at its end), [1] += accepted swaps
    unsigned long long round0;             // number of the launch's first round
    uint32_t n_rungs, swap_every, seed_lo, seed_hi;
    double jabs;
    long long n_bonds;
};

// one launch: `blocks` strips of `nw` waves (1 or 4) each; arguments as lat_strip_kernel
hipError_t strip_launch(bool pmj, int nw, unsigned blocks, size_t lds_bytes, hipStream_t stream, uint32_t *state, const LatGeom &g,
                        const StripArgs &a, uint64_t t0, uint32_t timesteps, const uint2 *keys, const LatThr *thr_steps,
                        uint32_t thr_stride, const LatThr *thr_replica, const uint32_t *jneg, uint32_t jneg_uniform,
                        unsigned long long *halo, unsigned long long *steps_out, const StripFinal &fin, const StripLadder &lad,
                        uint32_t n_replicas, uint32_t *err);

// resident workgroups per CU of the instantiation (pmj, nw, ladder) at `lds_bytes` of dynamic LDS: hipOccupancyMaxActiveBlocksPerMultiprocessor
int strip_blocks_per_cu(bool pmj, int nw, bool ladder, size_t lds_bytes);

} // namespace isingmc
