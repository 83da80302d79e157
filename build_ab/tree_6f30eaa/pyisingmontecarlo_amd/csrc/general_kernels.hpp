// General edge-list path: greedy-coloured independent sets, CSR gather, f64 local fields.
// Replaces the serial do_time_step loop (lattice.rs:204-207) for any graph the lattice recogniser
// rejects.  DESIGN.md S4; the CPU oracle (engine C) reproduces the configurations bit for bit, which
// is why every f64 operation below is written out (explicit fma, -ffp-contract=off).
//
// Layout: sites are renumbered colour-major ("positions"), each colour class padded to a multiple
// of 64 so that one wavefront covers 64 positions of ONE class = exactly two packed state words;
// the flip mask of a wave is its __ballot, written by lane 0 -- no atomics.
#pragma once
#include "philox.hpp"

namespace isingmc {

constexpr uint32_t PAD_SITE = 0xFFFFFFFFu;

struct GenGraphDev {
    const uint32_t *rowptr;  // n_pos + 1
    const uint32_t *nbr;     // neighbour POSITIONS
    const void *w;           // couplings, float (when lossless) or double
    const double *bias;      // per position, or nullptr (all zero)
    const uint32_t *site;    // original site id per position, PAD_SITE on padding
    const uint32_t *class_base; // n_colours + 1 positions (device copy of the colour-class boundaries)
    uint32_t n_colours;
    uint32_t n_pos;          // multiple of 64
    uint32_t n_words;        // n_pos / 32
};

// exp(x) for x = -beta dE, IEEE f64 ops + fma only (same bits as the oracle's orc_det_exp)
__device__ __forceinline__ double det_exp(double x)
{
    if (x >= 0.0) return 1.0;
    if (x < -40.0) return 0.0; // below 2^-53: can never beat a 53-bit uniform
    const double LOG2E = 1.4426950408889634074;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double kf = floor(fma(x, LOG2E, 0.5));
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const long long k = (long long)kf; // in [-58, 0]
    return p * __longlong_as_double((1023ll + k) << 52);
}

constexpr int GEN_ROW_BATCH = 8; // directed edges of a CSR row fetched together

// sum_e J_e s_q in adjacency order; the row is fetched in batches (see gen_sweep_kernel)
template <typename WT>
__device__ __forceinline__ double local_field(const GenGraphDev &G, const uint32_t *__restrict__ st,
                                              uint32_t p)
{
    const WT *w = static_cast<const WT *>(G.w);
    double field = 0.0;
    const uint32_t end = G.rowptr[p + 1];
    for (uint32_t e0 = G.rowptr[p]; e0 < end; e0 += GEN_ROW_BATCH) {
        uint32_t q[GEN_ROW_BATCH], word[GEN_ROW_BATCH];
        double j[GEN_ROW_BATCH];
#pragma unroll
        for (int i = 0; i < GEN_ROW_BATCH; i++) {
            const uint32_t e = min(e0 + i, end - 1);
            q[i] = G.nbr[e];
            j[i] = double(w[e]);
        }
#pragma unroll
        for (int i = 0; i < GEN_ROW_BATCH; i++) word[i] = st[q[i] >> 5];
#pragma unroll
        for (int i = 0; i < GEN_ROW_BATCH; i++)
            if (e0 + i < end) field += ((word[i] >> (q[i] & 31)) & 1u) ? j[i] : -j[i];
    }
    return field;
}

// One colour class of one timestep.  A thread owns one site for RB replicas (blockIdx.y = replica
// group): the CSR row (index + coupling per directed edge, the dominant stream) is read once per RB
// replicas instead of once per replica; only the bit gathers stay per replica.  The per-site
// arithmetic is replica-local and unchanged, so the result does not depend on RB.
template <typename WT, int RB>
__global__ __launch_bounds__(256) void gen_sweep_kernel(
    uint32_t *__restrict__ state, const GenGraphDev G, const uint32_t class_begin,
    const uint32_t class_end, const uint64_t t, const uint2 *__restrict__ keys,
    const double beta_uniform, const double *__restrict__ beta_replica, const uint32_t n_replicas)
{
    const uint32_t r0 = blockIdx.y * RB;
    const uint32_t p = class_begin + blockIdx.x * 256 + threadIdx.x;
    if (p >= class_end) return; // class sizes are multiples of 64: whole waves leave together
    const uint32_t site = G.site[p];
    const WT *w = static_cast<const WT *>(G.w);
    double field[RB];
#pragma unroll
    for (int k = 0; k < RB; k++) field[k] = 0.0;
    if (site != PAD_SITE) {
        // The row in batches of GEN_ROW_BATCH directed edges: all indices and couplings of a batch go out
        // together, then all state words of a replica -- two round trips per batch.  One edge per loop iteration
        // (index, then word) was two DEPENDENT round trips per edge: a degree-6 site waited twelve times, and
        // mid-size launches (too few waves to hide it) ran at a fifth of the big-graph rate.  The sum keeps the
        // adjacency order (f64 addition is not associative; the oracle adds in that order).
        const uint32_t end = G.rowptr[p + 1];
        for (uint32_t e0 = G.rowptr[p]; e0 < end; e0 += GEN_ROW_BATCH) {
            uint32_t q[GEN_ROW_BATCH];
            double j[GEN_ROW_BATCH];
#pragma unroll
            for (int i = 0; i < GEN_ROW_BATCH; i++) {
                const uint32_t e = min(e0 + i, end - 1); // clamped: a valid address; the term is skipped below
                q[i] = G.nbr[e];
                j[i] = double(w[e]);
            }
#pragma unroll
            for (int k = 0; k < RB; k++) {
                if (r0 + k >= n_replicas) break; // wave-uniform
                const uint32_t *st = state + size_t(r0 + k) * G.n_words;
                uint32_t word[GEN_ROW_BATCH];
#pragma unroll
                for (int i = 0; i < GEN_ROW_BATCH; i++) word[i] = st[q[i] >> 5];
#pragma unroll
                for (int i = 0; i < GEN_ROW_BATCH; i++)
                    if (e0 + i < end) field[k] += ((word[i] >> (q[i] & 31)) & 1u) ? j[i] : -j[i];
            }
        }
    }
    const double bias = (site != PAD_SITE && G.bias) ? G.bias[p] : 0.0;
#pragma unroll
    for (int k = 0; k < RB; k++) {
        const uint32_t r = r0 + k;
        if (r >= n_replicas) break; // wave-uniform
        uint32_t *st = state + size_t(r) * G.n_words;
        bool flip = false;
        if (site != PAD_SITE) {
            const double beta = beta_replica ? beta_replica[r] : beta_uniform;
            const double si = ((st[p >> 5] >> (p & 31)) & 1u) ? 1.0 : -1.0;
            const double dE = 2.0 * si * (bias - field[k]);
            flip = dE <= 0.0;
            if (!flip) {
                const uint4 rnd =
                    philox4x32_10(make_uint4(uint32_t(t), site >> 1, ctr2(t, 0, 0), DOM_GEN_SWEEP), keys[r]);
                const uint64_t x = (site & 1u) ? (uint64_t(rnd.w) << 32 | rnd.z) : (uint64_t(rnd.y) << 32 | rnd.x);
                const double u = double(x >> 11) * (1.0 / 9007199254740992.0);
                flip = u < det_exp(-beta * dE);
            }
        }
        const unsigned long long mask = __ballot(flip);
        if ((threadIdx.x & 63) == 0 && mask) {
            st[p >> 5] ^= uint32_t(mask);
            st[(p >> 5) + 1] ^= uint32_t(mask >> 32);
        }
    }
}

// random start: packed word w = Philox(key, (0, w>>2, 0, "GENI"))[w&3], padding bits cleared
__global__ __launch_bounds__(256) void gen_init_kernel(uint32_t *__restrict__ state,
                                                       const GenGraphDev G,
                                                       const uint2 *__restrict__ keys,
                                                       const uint32_t first_replica)
{
    const uint32_t r = first_replica + blockIdx.y;
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w >= G.n_words) return;
    const uint4 rnd = philox4x32_10(make_uint4(0, w >> 2, 0, DOM_GEN_INIT), keys[r]);
    uint32_t v = (w & 3) == 0 ? rnd.x : (w & 3) == 1 ? rnd.y : (w & 3) == 2 ? rnd.z : rnd.w;
    uint32_t valid = 0;
    for (int b = 0; b < 32; b++) valid |= uint32_t(G.site[32 * w + b] != PAD_SITE) << b;
    state[size_t(r) * G.n_words + w] = v & valid;
}

// per-block partial sums of E = sum_i s_i (field_i / 2 - h_i) and of M = sum_i s_i.
// partial_e[r][block], partial_m[r][block]; a second pass adds them in a fixed order, so the result
// is reproducible run to run (no floating-point atomics).
template <typename WT>
__global__ __launch_bounds__(256) void gen_measure_kernel(const uint32_t *__restrict__ state,
                                                          const GenGraphDev G,
                                                          double *__restrict__ partial_e,
                                                          long long *__restrict__ partial_m)
{
    __shared__ double se[4];
    __shared__ long long sm[4];
    const uint32_t r = blockIdx.y;
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    const uint32_t *st = state + size_t(r) * G.n_words;
    double e = 0.0;
    long long m = 0;
    if (p < G.n_pos && G.site[p] != PAD_SITE) {
        const double si = ((st[p >> 5] >> (p & 31)) & 1u) ? 1.0 : -1.0;
        const double field = local_field<WT>(G, st, p);
        e = si * (0.5 * field - (G.bias ? G.bias[p] : 0.0));
        m = (long long)si;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        e += __shfl_xor(e, off);
        m += __shfl_xor(m, off);
    }
    if ((threadIdx.x & 63) == 0) { se[threadIdx.x >> 6] = e; sm[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial_e[size_t(r) * gridDim.x + blockIdx.x] = (se[0] + se[1]) + (se[2] + se[3]);
        partial_m[size_t(r) * gridDim.x + blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    }
}

// fixed-order tree sum of one replica's partials: out_e[r], out_m[r]
__global__ __launch_bounds__(256) void gen_reduce_kernel(const double *__restrict__ partial_e,
                                                         const long long *__restrict__ partial_m,
                                                         const uint32_t n_partials,
                                                         double *__restrict__ out_e,
                                                         long long *__restrict__ out_m)
{
    __shared__ double se[256];
    __shared__ long long sm[256];
    const uint32_t r = blockIdx.x;
    double e = 0.0;
    long long m = 0;
    for (uint32_t i = threadIdx.x; i < n_partials; i += 256) {
        e += partial_e[size_t(r) * n_partials + i];
        m += partial_m[size_t(r) * n_partials + i];
    }
    se[threadIdx.x] = e;
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (int(threadIdx.x) < s) { se[threadIdx.x] += se[threadIdx.x + s]; sm[threadIdx.x] += sm[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out_e[r] = se[0]; out_m[r] = sm[0]; }
}

// ------------------------------------------------------------------------------------------------
// LDS-resident variant for small graphs (packed state <= GEN_RESIDENT_MAX_BYTES): one workgroup owns
// one replica for `timesteps` whole timesteps, all colour classes, with the spins in LDS and a
// workgroup barrier between classes -- instead of n_colours launches per timestep.  Same per-site
// arithmetic and Philox counters as gen_sweep_kernel => same configurations.
// energies_out (optional): E after every timestep, [replica][timesteps], reduced in a fixed order.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t GEN_RESIDENT_MAX_BYTES = 32 * 1024;

// LDS words the staged variant needs behind the state words (see gen_resident_kernel STAGE); edges2 = directed edges
// stage: 1 = everything, 2 = the topology only (row pointers, site ids, neighbour positions: the links of the dependent chain;
// couplings and biases stay in global memory, their addresses do not depend on loaded data)
__host__ __device__ inline uint32_t gen_stage_words(uint32_t n_pos, uint32_t edges2, bool has_bias, uint32_t w_bytes, int stage)
{
    uint32_t off = ((n_pos / 32) + 1u) & ~1u;             // state words, then 8-byte alignment
    if (stage == 1) {
        if (has_bias) off += 2 * n_pos;                    // f64 bias per position
        off += edges2 * (w_bytes / 4);                     // couplings (f32 or f64)
        off = (off + 1u) & ~1u;
    }
    return off + (n_pos + 1) + n_pos + edges2;             // rowptr, site, nbr
}

// STAGE: the graph itself (row pointers, neighbour positions, couplings, biases, site ids) is copied into LDS once per launch.
// A timestep of a small graph is a chain of DEPENDENT loads per colour class -- site -> row pointer -> neighbour -> spin word --
// and from global memory each link costs an L2 round trip although nothing but the spins ever changes: 2.2 us per timestep for
// 16 x 16 (c1).  Same arithmetic in the same order: bit-identical.
template <typename WT, int STAGE>
__global__ __launch_bounds__(1024) void gen_resident_kernel(
    uint32_t *__restrict__ state, const GenGraphDev G, const uint64_t t0, const uint32_t timesteps,
    const uint2 *__restrict__ keys, const double *__restrict__ beta_steps, const uint32_t beta_stride,
    const double *__restrict__ beta_replica, double *__restrict__ energies_out, const double self_energy, const uint32_t edges2)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t st[];
    __shared__ double red[16];
    const uint32_t r = blockIdx.x, tid = threadIdx.x, nthreads = blockDim.x;
    uint32_t *mine = state + size_t(r) * G.n_words;
    for (uint32_t i = tid; i < G.n_words; i += nthreads) st[i] = mine[i];
    const uint2 key = keys[r];
    GenGraphDev L = G;                                     // the arrays the loops below read: global, or their LDS copies
    const WT *w = static_cast<const WT *>(G.w);
    if constexpr (STAGE != 0) {
        uint32_t off = (G.n_words + 1u) & ~1u;
        if constexpr (STAGE == 1) {
            double *s_bias = reinterpret_cast<double *>(st + off);
            if (G.bias) off += 2 * G.n_pos;
            WT *s_w = reinterpret_cast<WT *>(st + off);
            off += edges2 * uint32_t(sizeof(WT) / 4);
            off = (off + 1u) & ~1u;
            if (G.bias)
                for (uint32_t i = tid; i < G.n_pos; i += nthreads) s_bias[i] = G.bias[i];
            for (uint32_t i = tid; i < edges2; i += nthreads) s_w[i] = w[i];
            L.bias = G.bias ? s_bias : nullptr;
            w = s_w;
        }
        uint32_t *s_rowptr = st + off;
        uint32_t *s_site = s_rowptr + G.n_pos + 1;
        uint32_t *s_nbr = s_site + G.n_pos;
        for (uint32_t i = tid; i <= G.n_pos; i += nthreads) s_rowptr[i] = G.rowptr[i];
        for (uint32_t i = tid; i < G.n_pos; i += nthreads) s_site[i] = G.site[i];
        for (uint32_t i = tid; i < edges2; i += nthreads) s_nbr[i] = G.nbr[i];
        L.rowptr = s_rowptr; L.site = s_site; L.nbr = s_nbr;
    }
    __syncthreads();
    for (uint32_t k = 0; k < timesteps; k++) {
        const uint64_t t = t0 + k;
        const double beta = beta_replica ? beta_replica[r] : beta_steps[size_t(k) * beta_stride];
        for (uint32_t c = 0; c < G.n_colours; c++) {
            const uint32_t begin = G.class_base[c], end = G.class_base[c + 1];
            for (uint32_t p = begin + tid; p < end; p += nthreads) { // whole waves: class sizes are multiples of 64
                const uint32_t site = L.site[p];
                bool flip = false;
                if (site != PAD_SITE) {
                    double field = 0.0;
                    for (uint32_t e = L.rowptr[p], ee = L.rowptr[p + 1]; e < ee; e++) {
                        const uint32_t q = L.nbr[e];
                        const double j = double(w[e]);
                        field += ((st[q >> 5] >> (q & 31)) & 1u) ? j : -j;
                    }
                    const double si = ((st[p >> 5] >> (p & 31)) & 1u) ? 1.0 : -1.0;
                    const double dE = 2.0 * si * ((L.bias ? L.bias[p] : 0.0) - field);
                    flip = dE <= 0.0;
                    if (!flip) {
                        const uint4 rnd = philox4x32_10(make_uint4(uint32_t(t), site >> 1, ctr2(t, 0, 0), DOM_GEN_SWEEP), key);
                        const uint64_t x = (site & 1u) ? (uint64_t(rnd.w) << 32 | rnd.z) : (uint64_t(rnd.y) << 32 | rnd.x);
                        const double u = double(x >> 11) * (1.0 / 9007199254740992.0);
                        flip = u < det_exp(-beta * dE);
                    }
                }
                const unsigned long long mask = __ballot(flip);
                if ((tid & 63) == 0 && mask) {
                    st[p >> 5] ^= uint32_t(mask);
                    st[(p >> 5) + 1] ^= uint32_t(mask >> 32);
                }
            }
            __syncthreads();
        }
        if (energies_out) {
            double e = 0.0;
            for (uint32_t p = tid; p < G.n_pos; p += nthreads) {
                if (L.site[p] == PAD_SITE) continue;
                double field = 0.0;
                for (uint32_t ed = L.rowptr[p], ee = L.rowptr[p + 1]; ed < ee; ed++) {
                    const uint32_t q = L.nbr[ed];
                    const double j = double(w[ed]);
                    field += ((st[q >> 5] >> (q & 31)) & 1u) ? j : -j;
                }
                const double si = ((st[p >> 5] >> (p & 31)) & 1u) ? 1.0 : -1.0;
                e += si * (0.5 * field - (L.bias ? L.bias[p] : 0.0));
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off);
            if ((tid & 63) == 0) red[tid >> 6] = e;
            __syncthreads();
            if (tid == 0) {
                double sum = 0.0;
                for (uint32_t wv = 0; wv < (nthreads + 63) / 64; wv++) sum += red[wv];
                energies_out[size_t(r) * timesteps + k] = sum + self_energy;
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < G.n_words; i += nthreads) mine[i] = st[i];
}

// ------------------------------------------------------------------------------------------------
// Parallel-tempering exchange round ON THE STREAM (host twin: pt_swap_round in
// host_logic.cpp -- same Philox counters, same det_exp, hence the same decisions).  One workgroup:
// the pairs (i, i+1) of the round's parity are disjoint, so one thread per pair swaps perm entries in
// place; then every rung relabels its slot's beta / thresholds if the slot is local.  No host sync.
// ------------------------------------------------------------------------------------------------
struct PtDev {
    const double *ladder;        // beta per rung
    const uint64_t *ladder_thr;  // per rung, host-computed: lattice {T3, T4} (exp of glibc); real-coupling path: one word = its RjBeta; else nullptr
    uint32_t thr_words;          // 64-bit words of ladder_thr per rung (2 / 1)
    uint32_t *perm;              // rung -> global slot
    const double *slot_energy;   // all slots (gathered)
    unsigned long long *counters; // [0] = round, [1] = total swaps
    uint32_t n_rungs, slot_offset, n_local;
    uint32_t seed_lo, seed_hi;
};

__global__ __launch_bounds__(1024) void pt_swap_kernel(const PtDev P, uint64_t *__restrict__ thr_local /*thr_words words per local slot*/,
                                                       double *__restrict__ beta_local, const uint32_t apply_only)
{
    const unsigned long long round = P.counters[0];
    __syncthreads(); // everyone has read the round before thread 0 bumps it
    unsigned swaps = 0;
    for (uint32_t i = uint32_t(round & 1) + 2 * threadIdx.x; !apply_only && i + 1 < P.n_rungs; i += 2 * blockDim.x) {
        const uint32_t sa = P.perm[i], sb = P.perm[i + 1];
        const double d = (P.ladder[i] - P.ladder[i + 1]) * (P.slot_energy[sa] - P.slot_energy[sb]);
        bool accept = d >= 0.0;
        if (!accept) {
            const uint4 rnd = philox4x32_10(make_uint4(i, uint32_t(round), uint32_t(round >> 32), 0x50545357u),
                                            make_uint2(P.seed_lo, P.seed_hi));
            const uint64_t x = (uint64_t(rnd.y) << 32) | rnd.x;
            accept = double(x >> 11) * (1.0 / 9007199254740992.0) < det_exp(d);
        }
        if (accept) {
            P.perm[i] = sb;
            P.perm[i + 1] = sa;
            swaps++;
        }
    }
    if (swaps) atomicAdd(&P.counters[1], (unsigned long long)swaps);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < P.n_rungs; i += blockDim.x) {
        const uint32_t slot = P.perm[i];
        if (slot >= P.slot_offset && slot < P.slot_offset + P.n_local) {
            const uint32_t r = slot - P.slot_offset;
            if (beta_local) beta_local[r] = P.ladder[i];
            if (P.ladder_thr)
                for (uint32_t k = 0; k < P.thr_words; k++) thr_local[size_t(P.thr_words) * r + k] = P.ladder_thr[size_t(P.thr_words) * i + k];
        }
    }
    if (threadIdx.x == 0 && !apply_only) P.counters[0] = round + 1;
}

} // namespace isingmc
