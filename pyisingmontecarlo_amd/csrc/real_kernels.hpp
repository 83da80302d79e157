// Replica-packed path for REAL couplings and arbitrary site biases: what the reference's f64 edge list
// (lattice.rs:46-50) and set_individual_bias / set_global_bias (lattice.rs:104-131, materialised at :186-189) can express and
// the bit-sliced kernels cannot -- Gaussian spin glasses, random fields, one biased site on an otherwise uniform lattice.
//
// Layout as the packed path (packed_kernels.hpp): colour-major positions, one 32-bit word per POSITION holding the spins
// of a group of 32 replicas.  One thread owns one position and decides its 32 replicas:
//   1. gather the own word and the neighbour words; a_e = own ^ neighbour_e (bit b: replica b's bond e is antiparallel);
//   2. the half energy change of a flip, X = s (hq - sum_e Jq_e s_e) = (s hq) - sum_e Jq_e + 2 sum_{e: a_e} Jq_e, depends on
//      the replica only through the index (a_0 .. a_{d-1}, own): its <= 32 values go into a per-thread column of LDS
//      (layout [entry][thread]: conflict-free), built with 15 + 32 integer adds per 32 replicas;
//   3. an 8 x 32 bit-matrix transposition (12 delta swaps of 4 instructions) turns the 8 words into 32 index bytes;
//   4. per replica: one 32-bit Philox word u (8 Philox4x32-10 calls per position = 4 attempts per call), Lambda_q(u) ~
//      -log2(u 2^-32) in Q24 from the bits of float(u) and a 2048-entry LDS table with linear interpolation, and the
//      integer test  max(X >> shift, 0) <= (Lambda_q * mant) >> 32  -- no exp, no f64, no 53-bit uniform per attempt.
// The CPU oracle (engine E, oracle/ising_oracle.c) takes the same decisions spin by spin from a direct integer field sum.
#pragma once
#include "philox.hpp"
#include "real_types.hpp"

namespace isingmc {

constexpr uint32_t RJ_PAD_SITE = 0xFFFFFFFFu; // == PAD_SITE (general_kernels.hpp)

// One step of the transposition: swap the high sub-blocks (bits m << s) of x with the low sub-blocks (bits m) of y.
__device__ __forceinline__ void rj_delta_swap(uint32_t &x, uint32_t &y, const uint32_t m, const int s)
{
    const uint32_t x2 = (m & x) | (~m & (y << s)); // v_lshlrev + v_bfi
    const uint32_t y2 = (m & (x >> s)) | (~m & y); // v_lshrrev + v_bfi
    x = x2;
    y = y2;
}

// w[c] bit b (c = index bit, b = replica)  ->  w[b & 7] byte (b >> 3) = the index of replica b (bit c = old w[c] bit b)
__device__ __forceinline__ void rj_transpose(uint32_t (&w)[8])
{
    rj_delta_swap(w[0], w[1], 0x55555555u, 1); rj_delta_swap(w[2], w[3], 0x55555555u, 1);
    rj_delta_swap(w[4], w[5], 0x55555555u, 1); rj_delta_swap(w[6], w[7], 0x55555555u, 1);
    rj_delta_swap(w[0], w[2], 0x33333333u, 2); rj_delta_swap(w[1], w[3], 0x33333333u, 2);
    rj_delta_swap(w[4], w[6], 0x33333333u, 2); rj_delta_swap(w[5], w[7], 0x33333333u, 2);
    rj_delta_swap(w[0], w[4], 0x0F0F0F0Fu, 4); rj_delta_swap(w[1], w[5], 0x0F0F0F0Fu, 4);
    rj_delta_swap(w[2], w[6], 0x0F0F0F0Fu, 4); rj_delta_swap(w[3], w[7], 0x0F0F0F0Fu, 4);
}

// Lambda_q(u): Q24 fixed point of 32 - log2(u) from the exponent field and the top mantissa bits of float(u) (round to
// nearest even), s_log[i] = {LT[i], LT[i+1] - LT[i]}; 159 << 24 for u = 0, 0 when u rounds up to 2^32
__device__ __forceinline__ uint32_t rj_lambda(const uint32_t u, const uint2 *s_log)
{
    const uint32_t bits = __float_as_uint(__uint2float_rn(u));
    const uint2 e = s_log[(bits >> 12) & 0x7FFu];
    const uint32_t val = e.x + __umulhi(e.y, bits << 20); // LT + ((D * low 12 mantissa bits) >> 12)
    return (159u << 24) - ((bits >> 23) << 24) - val;
}

// Index of a replica at a position: bit e < SLOTS = a_e (bond e antiparallel), bit SLOTS = the own spin (bit 4 when SLOTS == 4).
// SLOTS == 4: ONE table of 32 entries indexed by those 5 bits.  SLOTS == 7 / 11 / 15 / 23 / 31: the 8 / 12 / 16 / 24 / 32 index
// bits are cut into 2 / 3 / 4 / 6 / 8 nibbles with a 16-entry table each, X = the sum of the nibbles' entries (the constant
// - sum_e Jq_e and the own spin's +- hq live in the nibble that holds the own bit).
template <int SLOTS>
struct RjShape {
    static_assert(SLOTS == 4 || SLOTS == 7 || SLOTS == 11 || SLOTS == 15 || SLOTS == 23 || SLOTS == 31, "slots");
    static constexpr int NIB = SLOTS == 4 ? 0 : (SLOTS + 4) / 4;      // nibble tables (0: the single 32-entry table)
    static constexpr int ENTRIES = SLOTS == 4 ? 32 : 16 * NIB;        // LDS words per thread
    static constexpr int WORDS = SLOTS <= 7 ? 8 : SLOTS <= 15 ? 16 : SLOTS <= 23 ? 24 : 32; // index words before transposition
    static constexpr int OWN_BIT = SLOTS == 4 ? 4 : SLOTS;
    static constexpr int THREADS = SLOTS <= 7 ? 256 : SLOTS <= 15 ? 128 : 64; // workgroup size (48 / 64, 96 / 128 table words per thread)
};

// The per-thread column of X values.  HSCALE: 1 for the sweep (X), 2 for the measurement of a general graph
// (X + s hq = 2 s hq - SJ + 2 sum_{a} Jq).  FOLD: store max(X >> shift, 0) (one beta for all replicas, SLOTS == 4) instead of X.
template <int SLOTS, int HSCALE, bool FOLD>
__device__ __forceinline__ void rj_build_tables(uint32_t *s_x, const uint32_t tid, const int32_t (&jq)[SLOTS], const int32_t hq,
                                                const uint32_t shift)
{
    // (unsigned arithmetic throughout: 2 x a subset sum can pass 2^31 on the way to an X that fits int32 again -- wrap-around is
    //  exact modulo 2^32, signed overflow would be undefined)
    using SH = RjShape<SLOTS>;
    uint32_t sj = 0;
#pragma unroll
    for (int e = 0; e < SLOTS; e++) sj += uint32_t(jq[e]);
    const uint32_t hs = uint32_t(HSCALE) * uint32_t(hq);
    if constexpr (SLOTS == 4) {
        uint32_t sub[16]; // 2 x subset sums of the four couplings
        sub[0] = 0;
#pragma unroll
        for (int n = 1; n < 16; n++) sub[n] = sub[n & (n - 1)] + 2u * uint32_t(jq[__builtin_ctz(n)]);
#pragma unroll
        for (int n = 0; n < 16; n++) {
            const int32_t x0 = int32_t(sub[n] - sj - hs), x1 = int32_t(sub[n] - sj + hs); // own spin down / up
            if constexpr (FOLD) {
                s_x[n * RjShape<SLOTS>::THREADS + tid] = uint32_t(max(x0 >> shift, 0));
                s_x[(16 + n) * RjShape<SLOTS>::THREADS + tid] = uint32_t(max(x1 >> shift, 0));
            } else {
                s_x[n * RjShape<SLOTS>::THREADS + tid] = uint32_t(x0);
                s_x[(16 + n) * RjShape<SLOTS>::THREADS + tid] = uint32_t(x1);
            }
        }
    } else {
#pragma unroll
        for (int nb = 0; nb < SH::NIB; nb++) {
            uint32_t sub[16];
            sub[0] = nb == SH::OWN_BIT / 4 ? 0u - sj - hs : 0u; // the own bit's nibble carries the constants (own spin down)
#pragma unroll
            for (int n = 1; n < 16; n++) {
                const int bit = 4 * nb + __builtin_ctz(n); // the index bit this pattern adds
                const uint32_t add = bit < SLOTS ? 2u * uint32_t(jq[bit < SLOTS ? bit : 0]) : bit == SH::OWN_BIT ? 2u * hs : 0u;
                sub[n] = sub[n & (n - 1)] + add;
            }
#pragma unroll
            for (int n = 0; n < 16; n++) s_x[(16 * nb + n) * RjShape<SLOTS>::THREADS + tid] = sub[n];
        }
    }
}

// X (or the folded threshold operand) of the replica whose index is `idx` (one byte per transposition, low byte first)
template <int SLOTS>
__device__ __forceinline__ uint32_t rj_lookup(const uint32_t *s_x, const uint32_t tid, const uint32_t idx)
{
    using SH = RjShape<SLOTS>;
    if constexpr (SLOTS == 4) return s_x[idx * RjShape<SLOTS>::THREADS + tid];
    else {
        uint32_t x = s_x[(idx & 15u) * RjShape<SLOTS>::THREADS + tid];
#pragma unroll
        for (int nb = 1; nb < SH::NIB; nb++) x += s_x[(16u * nb + ((idx >> (4 * nb)) & 15u)) * RjShape<SLOTS>::THREADS + tid];
        return x;
    }
}

// gather: own word, neighbour words, couplings, bias of position p; w[] = the index words before transposition
template <int SLOTS>
__device__ __forceinline__ void rj_gather(const uint32_t *__restrict__ st, const RjGraphDev &G, const uint32_t p, uint32_t &own,
                                          int32_t (&jq)[SLOTS], int32_t &hq, uint32_t (&w)[RjShape<SLOTS>::WORDS])
{
    uint32_t q[SLOTS];
    own = st[p];
#pragma unroll
    for (int e = 0; e < SLOTS; e++) q[e] = G.nbr[size_t(e) * G.n_pos + p];
#pragma unroll
    for (int e = 0; e < SLOTS; e++) jq[e] = G.jq[size_t(e) * G.n_pos + p];
    hq = G.hq[p];
#pragma unroll
    for (int e = 0; e < RjShape<SLOTS>::WORDS; e++) w[e] = 0u;
#pragma unroll
    for (int e = 0; e < SLOTS; e++) w[e] = own ^ st[q[e]];
    w[RjShape<SLOTS>::OWN_BIT] = own;
}

// a site without bonds (class 1 of a two-class graph in the measurement): only the own word and the bias
template <int SLOTS>
__device__ __forceinline__ void rj_gather_bias_only(const uint32_t *__restrict__ st, const RjGraphDev &G, const uint32_t p, uint32_t &own,
                                                    int32_t (&jq)[SLOTS], int32_t &hq, uint32_t (&w)[RjShape<SLOTS>::WORDS])
{
    own = st[p];
    hq = G.hq[p];
#pragma unroll
    for (int e = 0; e < SLOTS; e++) jq[e] = 0;
#pragma unroll
    for (int e = 0; e < RjShape<SLOTS>::WORDS; e++) w[e] = 0u;
    w[RjShape<SLOTS>::OWN_BIT] = own;
}

// index words -> index bytes: w[8 G + (b & 7)] byte (b >> 3) = bits 8 G .. 8 G + 7 of replica b's index
template <int SLOTS>
__device__ __forceinline__ void rj_transpose_all(uint32_t (&w)[RjShape<SLOTS>::WORDS])
{
#pragma unroll
    for (int G = 0; G < RjShape<SLOTS>::WORDS / 8; G++) {
        uint32_t x[8];
#pragma unroll
        for (int e = 0; e < 8; e++) x[e] = w[8 * G + e];
        rj_transpose(x);
#pragma unroll
        for (int e = 0; e < 8; e++) w[8 * G + e] = x[e];
    }
}

template <int SLOTS>
__device__ __forceinline__ uint32_t rj_index(const uint32_t (&w)[RjShape<SLOTS>::WORDS], const int b)
{
    uint32_t idx = (w[b & 7] >> (8 * (b >> 3))) & 0xFFu;
#pragma unroll
    for (int G = 1; G < RjShape<SLOTS>::WORDS / 8; G++) idx |= ((w[8 * G + (b & 7)] >> (8 * (b >> 3))) & 0xFFu) << (8 * G);
    return idx;
}

// one colour class of one timestep; blockIdx.y = replica group; a workgroup walks 256-position blocks of the class
// PARTIAL: a group of which this container owns only the replica bits 4 q_lo .. 4 q_hi - 1 (few experiments; the first / last
// group of a shard): only the Philox calls q_lo .. q_hi - 1 are drawn and only their replicas decided -- a replica's decisions
// depend on nothing but its own spins, beta and bit position (S7), so the bits outside are nobody's business and stay as they are.
// The cost of a position then is ~250 vector instructions + ~21 per decided replica instead of 932.
// HEAVY: the graph has sites that quantise at a coarser scale of their own (G.dshift; spec S7: X is shifted right by less, the
// acceptance bound by the rest of d_p) -- one more shift per attempt.
template <int SLOTS, bool UB, bool PARTIAL = false, bool HEAVY = PARTIAL>
__global__ __launch_bounds__(RjShape<SLOTS>::THREADS) void rj_sweep_kernel(uint32_t *__restrict__ state, const RjGraphDev G, const uint32_t class_begin,
                                                              const uint32_t real_end, const uint64_t t,
                                                              const uint2 *__restrict__ group_keys, const RjBeta *__restrict__ betas,
                                                              const uint32_t q_lo, const uint32_t q_hi)
{
    __shared__ uint2 s_log[RJ_LOG_INTERVALS];
    __shared__ uint32_t s_x[RjShape<SLOTS>::ENTRIES * RjShape<SLOTS>::THREADS];
    const uint32_t tid = threadIdx.x, g = blockIdx.y;
#pragma unroll
    for (int i = 0; i < RJ_LOG_INTERVALS / RjShape<SLOTS>::THREADS; i++) s_log[tid + RjShape<SLOTS>::THREADS * i] = G.logtab[tid + RjShape<SLOTS>::THREADS * i];
    __syncthreads();
    uint32_t *st = state + size_t(g) * G.n_pos;
    const uint2 key = group_keys[g];
    const PhiloxVKeys vk = philox_vkeys(key);
    const RjBeta *gb = betas + (UB ? 0 : size_t(32) * g);
    constexpr bool FOLD = UB && SLOTS == 4;
    const uint32_t shift0 = gb[0].shift, mant0 = gb[0].mant;

    for (uint32_t base = class_begin + blockIdx.x * RjShape<SLOTS>::THREADS; base < real_end; base += gridDim.x * RjShape<SLOTS>::THREADS) {
        const uint32_t p = base + tid;
        if (p >= real_end) continue; // (no barrier below: a thread reads only its own column of s_x)
        uint32_t own, w[RjShape<SLOTS>::WORDS];
        int32_t jq[SLOTS], hq;
        rj_gather<SLOTS>(st, G, p, own, jq, hq, w);
        // heavy site: X is in units of 2^(k + dsh); of those dsh binary places min(shift, dsh) come off the right shift of X, the
        // rest off the bound (spec S7)
        uint32_t dsh = 0, sx0 = shift0, sy0 = 0;
        if constexpr (HEAVY) {
            dsh = G.dshift ? uint32_t(G.dshift[p]) : 0u;
            const uint32_t m = min(shift0, dsh);
            sx0 = shift0 - m;
            sy0 = dsh - m;
        }
        rj_build_tables<SLOTS, 1, FOLD>(s_x, tid, jq, hq, sx0);
        rj_transpose_all<SLOTS>(w);
        uint32_t flips = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { // Philox call j serves replica bits 4j .. 4j+3
            if constexpr (PARTIAL) {
                if (uint32_t(j) < q_lo || uint32_t(j) >= q_hi) continue; // uniform over the launch
            }
            const uint4 rnd = philox4x32_10(make_uint4(uint32_t(t), p, DOM_RJ_SWEEP, ctr2(t, 0, uint32_t(j))), key, vk);
            const uint32_t u4[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int b = 4 * j + i;
                const uint32_t xv = rj_lookup<SLOTS>(s_x, tid, rj_index<SLOTS>(w, b));
                uint32_t shift = UB ? sx0 : gb[b].shift, sy = sy0;
                const uint32_t mant = UB ? mant0 : gb[b].mant;
                if constexpr (HEAVY && !UB) {
                    const uint32_t m = min(shift, dsh);
                    shift -= m;
                    sy = dsh - m;
                }
                const uint32_t xpos = FOLD ? xv : uint32_t(max(int32_t(xv) >> shift, 0));
                uint32_t y = __umulhi(rj_lambda(u4[i], s_log), mant);
                if constexpr (HEAVY) y >>= sy;
                flips |= uint32_t(xpos <= y) << b;
            }
        }
        st[p] = own ^ flips;
    }
}

// -2 x (energy in units of 2^k) and the up spins, per replica.  General graphs: sum over the real positions of
// X + s hq = 2 s hq - s F (every bond is seen from both ends).  BIP (two colour classes: every bond joins class 0 to
// class 1): the bonds are taken from the class-0 positions only (p < class0_end: 2 X = 2 s hq - 2 s F); the class-1
// positions add just their bias term 2 s hq (as sites without bonds: no gathers) and are not visited at all when the graph
// has no biases (scan_end = class0_end; with UP the scan covers every position).  A thread walks positions
// blockIdx.x * 256 + tid, + gridDim.x * 256, ... with 32 int64 accumulators; one wave reduction per replica at the end.
// UP: also count the up spins (get_magnetisations); the energy-only callers save 32 registers per thread (a wave more per SIMD)
template <int SLOTS, bool BIP, bool UP>
__global__ __launch_bounds__(RjShape<SLOTS>::THREADS) void rj_measure_kernel(const uint32_t *__restrict__ state, const RjGraphDev G,
                                                                const uint32_t *__restrict__ site, const uint32_t class0_end,
                                                                const uint32_t scan_end, unsigned long long *__restrict__ out)
{
    __shared__ uint32_t s_x[RjShape<SLOTS>::ENTRIES * RjShape<SLOTS>::THREADS];
    const uint32_t tid = threadIdx.x, g = blockIdx.y;
    const uint32_t *st = state + size_t(g) * G.n_pos;
    long long acc[32];
    uint32_t up[UP ? 32 : 1];
#pragma unroll
    for (int b = 0; b < 32; b++) acc[b] = 0;
#pragma unroll
    for (int b = 0; b < (UP ? 32 : 1); b++) up[b] = 0;
    for (uint32_t base = blockIdx.x * RjShape<SLOTS>::THREADS; base < scan_end; base += gridDim.x * RjShape<SLOTS>::THREADS) { // base: wave-uniform
        const uint32_t p = base + tid;
        if (site[p] == RJ_PAD_SITE) continue;
        uint32_t own, w[RjShape<SLOTS>::WORDS];
        int32_t jq[SLOTS], hq;
        // class boundaries are multiples of 256: uniform per workgroup.  Class 1 of a two-class graph: only the bias term
        // 2 s hq = 2 X of a site without bonds -- no gathers
        if (!BIP || base < class0_end) rj_gather<SLOTS>(st, G, p, own, jq, hq, w);
        else rj_gather_bias_only<SLOTS>(st, G, p, own, jq, hq, w);
        rj_build_tables<SLOTS, BIP ? 1 : 2, false>(s_x, tid, jq, hq, 0u);
        rj_transpose_all<SLOTS>(w);
#pragma unroll
        for (int b = 0; b < 32; b++) {
            const long long x = (long long)int32_t(rj_lookup<SLOTS>(s_x, tid, rj_index<SLOTS>(w, b)));
            acc[b] += BIP ? 2 * x : x;
            if constexpr (UP) up[b] += (own >> b) & 1u;
        }
    }
    // Reduce-scatter over the wave: in step k (lane distance 1, 2, 4, 8, 16) a lane keeps the half of its sums whose replica
    // index has bit k equal to its own lane bit and hands the other half to its partner -- 16 + 8 + 4 + 2 + 1 exchanges instead
    // of 32 full butterflies; lane l then holds replica l & 31, summed over the lanes that agree with it in bit 5, and one
    // more exchange (distance 32) finishes.  Four waves meet in LDS: 32 atomics per workgroup, all to different addresses
    // (one atomic per wave and replica took longer than the scan itself: 164 000 atomics on 128 addresses).
    const uint32_t lane = tid & 63u;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const bool hi = (lane >> k) & 1u;
        const int half = 16 >> k; // sums that survive this step
#pragma unroll
        for (int i = 0; i < half; i++) {
            // before step k the surviving sums sit at acc[0 .. 2 half); entry j stands for a replica index whose bit k is j / half
            const long long keep = hi ? acc[i + half] : acc[i], send = hi ? acc[i] : acc[i + half];
            acc[i] = keep + __shfl_xor(send, 1 << k);
            if constexpr (UP) {
                const uint32_t ukeep = hi ? up[i + half] : up[i], usend = hi ? up[i] : up[i + half];
                up[i] = ukeep + __shfl_xor(usend, 1 << k);
            }
        }
    }
    long long a = acc[0] + __shfl_xor(acc[0], 32);
    uint32_t u = UP ? up[0] + __shfl_xor(up[0], 32) : 0u;
    // which replica does lane l hold?  Step k kept, of the two halves, the one matching lane bit k, and the halves were split by
    // the TOP remaining index bit: step 0 decided index bit 4, step 1 bit 3, ... step 4 bit 0
    const uint32_t b = ((lane & 1u) << 4) | ((lane & 2u) << 2) | (lane & 4u) | ((lane & 8u) >> 2) | ((lane & 16u) >> 4);
    constexpr int NWAVES = RjShape<SLOTS>::THREADS / 64;
    __shared__ long long red_a[NWAVES][32];
    __shared__ uint32_t red_u[NWAVES][32];
    if (lane < 32) { red_a[tid >> 6][b] = a; red_u[tid >> 6][b] = u; }
    __syncthreads();
    if (tid < 32) {
        long long ta = 0;
        uint32_t tu = 0;
#pragma unroll
        for (int wv = 0; wv < NWAVES; wv++) { ta += red_a[wv][tid]; tu += red_u[wv][tid]; }
        if (ta != 0) atomicAdd(out + 2 * (size_t(32) * g + tid), (unsigned long long)ta);
        if (UP && tu != 0) atomicAdd(out + 2 * (size_t(32) * g + tid) + 1, (unsigned long long)tu);
    }
}

// tempering on the stream: energies of the local slots from the two measurement counters of a slot (hi level, lo level: each
// -2 x an exact integer sum), E = (2^kE hi + 2^(kE - 24) lo) + self-loop constant -- the arithmetic of the host's pk_energy: the
// same bits; the counters are left as they are (the next measurement zeroes them)
__attribute__((unused)) static __global__ void rj_energy_from_counts_kernel(unsigned long long *__restrict__ meas, const uint32_t first_slot,
                                                                            const uint32_t n, const int k_energy, const double self_energy,
                                                                            double *__restrict__ out)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) {
        const long long hi = (long long)meas[2 * size_t(first_slot + r)], lo = (long long)meas[2 * size_t(first_slot + r) + 1];
        out[r] = (ldexp(double(-(hi / 2)), k_energy) + ldexp(double(-(lo / 2)), k_energy - 24)) + self_energy;
    }
}

} // namespace isingmc
