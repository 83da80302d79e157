// Host-visible types and the launcher of the persistent strip kernel (strip_kernels.hpp).  The kernel lives in a
// translation unit of its own (strip_kernels.hip): instantiated next to the streaming kernels in isingmc.hip it
// changed THEIR register allocation (lat_sweep_loop_kernel went from 64 to 66 VGPRs = one wave per SIMD less).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace isingmc {

struct LatGeom;
struct LatThr;

constexpr uint32_t STRIP_MAX_WAVES_PER_CU = 16;           // residency by grid size alone (Guideline 16: <= 4 blocks of 256 threads)
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

// one launch: `blocks` strips of `nw` waves (1 or 4) each; arguments as lat_strip_kernel
hipError_t strip_launch(bool pmj, int nw, unsigned blocks, size_t lds_bytes, hipStream_t stream, uint32_t *state, const LatGeom &g,
                        const StripArgs &a, uint64_t t0, uint32_t timesteps, const uint2 *keys, const LatThr *thr_steps,
                        uint32_t thr_stride, const LatThr *thr_replica, const uint32_t *jneg, uint32_t jneg_uniform,
                        unsigned long long *halo, unsigned long long *steps_out, const StripFinal &fin, uint32_t n_replicas,
                        uint32_t *err);

} // namespace isingmc
