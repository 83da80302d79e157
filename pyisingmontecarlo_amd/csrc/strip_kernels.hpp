// Persistent strip kernel for mid-size lattices: ALL timesteps of a call in ONE launch, with no grid-wide
// synchronisation (replaces the two dependent launches per timestep that bound launches of a few waves per SIMD:
// the 1024^2 x 64-rung tempering ladder of BASELINE config c3 spent 11 of its 18 us per timestep between kernels).
//
// A replica is cut into horizontal strips of S rows, S x (quads per row) = 64 NW: one workgroup of NW waves owns one
// strip for the whole launch and keeps both colour planes of it in LDS (2 KB per wave + two halo rows).  A
// half-sweep of colour c reads the other colour's rows y-1 / y+1, so a strip needs exactly one row of each vertical
// neighbour strip per half-sweep.  Those two rows travel through global memory as 8-byte {tag, word} granules, each
// ONE relaxed agent-scope atomic store / load (cdna_hip_programming.md Guideline 16, form R2: the datum is its own
// flag, no fence, no separate flag, correct for any workgroup -> CU / XCD placement): after its half-sweep j a
// workgroup stores its top and bottom row with tag = epoch + j + 1; before half-sweep j + 1 its neighbours poll
// those granules until the tag matches.  Nothing else is shared during the launch: the strips' spins are loaded
// from the state array at the start (the first half-sweep's halo rows too: they were written before the launch)
// and stored back at the end.  A workgroup cannot overwrite a granule its neighbour still needs: it can only be
// one half-sweep ahead of it, and the colours alternate.
//
// NW = 1: a strip is ONE wavefront -- no workgroup barrier anywhere (a wave's LDS accesses execute in order), every
// wave waits only for its own two neighbours, and the four waves of a SIMD drift apart, so one wave's hand-off
// latency is covered by the others' arithmetic.  NW = 4: 256-thread workgroups, one barrier per half-sweep.
//
// Every workgroup of the launch must be resident at once (a strip spins on its neighbours): the host launches at
// most STRIP_MAX_WAVES_PER_CU x #CUs waves and serialises strip launches of one process on a device; every spin
// is bounded by the constant-rate counter (s_memrealtime) -- on a timeout the workgroup raises *err and leaves, and
// its neighbours follow within one poll.  Same Philox counters (global quad index, timestep, colour) as the
// per-colour launches => bit-identical configurations.
#pragma once
#include "lattice_kernels.hpp"
#include "strip_types.hpp"

namespace isingmc {

// exp(x) for x <= 0 with IEEE f64 ops + fma only: the device twin of det_exp (general_kernels.hpp) / orc_det_exp
__device__ __forceinline__ double strip_det_exp(double x)
{
    if (x >= 0.0) return 1.0;
    if (x < -40.0) return 0.0;
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

// bounded wait for a granule {tag, value}: false on a timeout (or when another workgroup has raised *err)
__device__ __forceinline__ bool strip_wait_granule(const unsigned long long *p, const uint32_t want, uint32_t &value, uint32_t *err)
{
    const unsigned long long start = __builtin_amdgcn_s_memrealtime();
    uint32_t spins = 0;
    for (;;) {
        const unsigned long long v = __hip_atomic_load((strip_gu64)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (uint32_t(v >> 32) == want) {
            value = uint32_t(v);
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0 && (__builtin_amdgcn_s_memrealtime() - start > STRIP_TIMEOUT_TICKS ||
                                     __hip_atomic_load((strip_gu32)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            atomicOr(err, STRIP_ERR_TIMEOUT);
            return false;
        }
    }
}

// granules of one replica: [strip][plane][side: 0 = its top row, 1 = its bottom row][wpr]
__device__ __forceinline__ size_t strip_granule(const LatGeom &g, uint32_t strip, uint32_t plane, uint32_t side)
{
    return ((size_t(strip) * 2 + plane) * 2 + side) * g.wpr;
}

template <bool PMJ>
__device__ __forceinline__ void strip_update_quad(uint32_t *lds, const uint32_t PL, const LatGeom &g, const uint32_t colour,
                                                  const uint32_t yl, const uint32_t col, const uint32_t Q, const bool odd, const uint64_t t,
                                                  const uint2 key, const PhiloxVKeys &vk, const LatThr thr,
                                                  const uint32_t *__restrict__ jn, const uint32_t jneg_uniform, const QuadRandom &R,
                                                  uint32_t new_words[4], const bool measure, uint32_t &sat, uint32_t &up)
{
    const uint32_t wpr = g.wpr, xw = 4 * col;
    uint32_t *ownp = lds + colour * PL;
    const uint32_t *othp = lds + (1 - colour) * PL;
    const uint32_t w0 = yl * wpr + xw;
    QuadSigns js;
    load_signs<PMJ>(jn, g, Q, js); // +-J: sign words from global memory (shared by all replicas, L2-resident), issued first
    const uint4 o4 = *reinterpret_cast<const uint4 *>(ownp + w0);
    const uint4 c4 = *reinterpret_cast<const uint4 *>(othp + w0);
    const uint4 u4 = *reinterpret_cast<const uint4 *>(othp + w0 - wpr);
    const uint4 d4 = *reinterpret_cast<const uint4 *>(othp + w0 + wpr);
    const uint32_t sx = odd ? (xw + 4 == wpr ? 0 : xw + 4) : (xw == 0 ? wpr : xw) - 1;
    QuadNbr n;
    n.si[0] = othp[yl * wpr + sx];
    uint32_t own[4] = {o4.x, o4.y, o4.z, o4.w}, acc[4];
    n.ce[0] = c4.x; n.ce[1] = c4.y; n.ce[2] = c4.z; n.ce[3] = c4.w;
    n.up[0] = u4.x; n.up[1] = u4.y; n.up[2] = u4.z; n.up[3] = u4.w;
    n.dn[0] = d4.x; n.dn[1] = d4.y; n.dn[2] = d4.z; n.dn[3] = d4.w;
    side_words(n, odd);
    const uint32_t widx[4] = {0, 0, 0, 0}; // only the PMJ = false bond masks take it, and ignore it
#ifdef ISINGMC_STRIP_NO_PRECOMPUTE // A/B build: the random words drawn after the wait, as in the streaming kernels
    (void)R;
    quad_flips<PMJ>(own, n, widx, g, colour, t, key, vk, thr, js, jneg_uniform, Q, acc);
#else
    quad_flips_pre<PMJ>(own, n, widx, g, colour, t, key, vk, thr, js, jneg_uniform, Q, R, acc);
#endif
#pragma unroll
    for (int q = 0; q < 4; q++) new_words[q] = own[q] ^ acc[q];
    *reinterpret_cast<uint4 *>(ownp + w0) = make_uint4(new_words[0], new_words[1], new_words[2], new_words[3]);
    if (measure) quad_measure<PMJ>(new_words, n, js, jneg_uniform, sat, up); // wave-uniform
}

// grid: n_replicas * n_strips workgroups of 64 NW threads; dynamic LDS: 2 planes x (S + 2) rows x wpr words + 16 words
// thr_steps / thr_stride / thr_replica: as lat_resident_kernel.  steps_out (optional): satisfied bonds / up spins
// after every timestep, [step][replica][2], zeroed by the host (the strips of a replica add into it).
// fin (optional): energies of the final configurations, see StripFinal.
// LAD: exchange rounds inside the launch (StripLadder); a separate instantiation, so that launches without them pay nothing
template <bool PMJ, int NW, bool LAD>
__global__ __launch_bounds__(64 * NW, 4) void lat_strip_kernel(
    uint32_t *__restrict__ state, const LatGeom g, const StripArgs a, const uint64_t t0, const uint32_t timesteps,
    const uint2 *__restrict__ keys, const LatThr *__restrict__ thr_steps, const uint32_t thr_stride,
    const LatThr *__restrict__ thr_replica, const uint32_t *__restrict__ jneg, const uint32_t jneg_uniform,
    unsigned long long *__restrict__ halo, unsigned long long *__restrict__ steps_out, const StripFinal fin, const StripLadder lad,
    const uint32_t n_replicas, uint32_t *__restrict__ err)
{
    constexpr uint32_t NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[]; // [plane][S + 2 rows][wpr] + 16 words of scratch
    const uint32_t tid = threadIdx.x, wpr = g.wpr, S = a.S;
    const uint32_t PL = (S + 2) * wpr;
    uint32_t *red = lds + 2 * PL; // NW > 1: [0..7] per-wave partial sums, [8] bail flag
    uint32_t r, strip;
    if (a.xcd_remap) { // blocks b and b + 8 share an XCD: keep a replica's strips (which exchange rows) on one L2
        const uint32_t x = blockIdx.x & 7u, m = blockIdx.x >> 3;
        strip = m % a.n_strips;
        r = x + 8 * (m / a.n_strips);
    } else {
        r = blockIdx.x / a.n_strips;
        strip = blockIdx.x - r * a.n_strips;
    }
    const uint32_t y0 = strip * S;
    const uint32_t up_strip = strip == 0 ? a.n_strips - 1 : strip - 1, dn_strip = strip + 1 == a.n_strips ? 0 : strip + 1;
    uint32_t *mine = state + size_t(r) * 2 * g.wpp;
    unsigned long long *rep_halo = halo + size_t(r) * a.n_strips * 4 * wpr;

    // ---- load: own rows (S * wpr = 4 NT words per plane: one 16-byte load per thread and plane) + the two halo rows
    //      of both planes, all written before this launch
#pragma unroll
    for (uint32_t p = 0; p < 2; p++)
        reinterpret_cast<uint4 *>(lds + p * PL + wpr)[tid] = reinterpret_cast<const uint4 *>(mine + size_t(p) * g.wpp + size_t(y0) * wpr)[tid];
    const uint32_t y_up = (y0 == 0 ? g.H : y0) - 1, y_dn = (y0 + S == g.H) ? 0 : y0 + S;
    for (uint32_t i = tid; i < 4 * wpr; i += NT) {
        const uint32_t p = i / (2 * wpr), side = (i / wpr) & 1u, w = i % wpr;
        lds[p * PL + (side ? (S + 1) * wpr : 0) + w] = mine[size_t(p) * g.wpp + size_t(side ? y_dn : y_up) * wpr + w];
    }
    if (NW > 1 && tid == 0) red[8] = 0;
    // in-kernel tempering: the rung this replica holds (the inverse of perm), found by the workgroup's threads
    uint32_t my_rung = 0;
    if constexpr (LAD) {
        if (tid == 0) red[9] = 0;
        if constexpr (NW > 1) __syncthreads();
        for (uint32_t i = tid; i < lad.n_rungs; i += NT)
            if (lad.perm_in[i] == r) red[9] = i; // exactly one hit (perm is a permutation of the slots; single shard: slot = replica)
        if constexpr (NW > 1) __syncthreads();
        my_rung = __builtin_amdgcn_readfirstlane(red[9]);
    }
    const uint2 key = keys[r];
    const PhiloxVKeys vk = philox_vkeys(key);

    // thread -> quad of the strip.  NW = 4: a pair of waves shares 2 * rpw consecutive rows, wave 0 the even ones, wave 1
    // the odd ones (rpw = 64 / quads per row), so the row parity is uniform per wavefront (as thread_to_quad<true>).
    // NW = 1: the wave's 64 quads in row-major order, parity per lane.
    const uint32_t ql = a.qpr_log2, wave = tid >> 6, lane = tid & 63u;
    uint32_t yrel, col;
    if (NW == 1 || ql >= 6) {
        yrel = tid >> ql;
        col = tid & ((1u << ql) - 1);
    } else {
        yrel = ((wave >> 1) << (7 - ql)) + 2 * (lane >> ql) + (wave & 1u);
        col = lane & ((1u << ql) - 1);
    }
    const uint32_t yl = yrel + 1, y_global = y0 + yrel;
    const bool top_row = yrel == 0, bottom_row = yrel + 1 == S;
    const uint32_t Q = y_global * (wpr >> 2) + col; // the GLOBAL quad index: the Philox counter of the per-colour launches
    // The random words of a half-sweep do not depend on the spins: they are drawn BEFORE the wait for the neighbour
    // strips' rows (7 + 1 Philox calls = two thirds of a half-sweep's work), so that the hand-off latency of the
    // granules (~1-2 us) is covered by work instead of adding to every half-sweep.
    QuadRandom R;
#ifndef ISINGMC_STRIP_NO_PRECOMPUTE
    quad_random(R, Q, 0, t0, key, vk);
#endif

    for (uint32_t k = 0; k < timesteps; k++) {
        LatThr thr;
        if constexpr (LAD) thr = LatThr{lad.ladder_thr[2 * size_t(my_rung)], lad.ladder_thr[2 * size_t(my_rung) + 1]};
        else thr = thr_replica ? thr_replica[r] : thr_steps[size_t(k) * thr_stride];
        uint32_t sat = 0, up = 0;
        const bool last_step = k + 1 == timesteps;
        const bool exchange = LAD && !last_step && (k + 1) % lad.swap_every == 0; // an exchange round follows this timestep
        const bool measure = (steps_out != nullptr) || (fin.counts != nullptr && last_step) || exchange;
#pragma unroll 1
        for (uint32_t colour = 0; colour < 2; colour++) {
            const uint32_t j = 2 * k + colour; // half-sweep index of this launch
#ifndef ISINGMC_STRIP_NO_PRIO
            // The wait for the neighbours' rows, the update and the publication of the own boundary rows are on the chain of hand-offs
            // between neighbouring strips; the draw of the next half-sweep's random words (half of the arithmetic) is not: the waves
            // of a SIMD that are on the chain go first (s_setprio: -11 % per timestep at 1024^2 x 64, profiles/r04_strip_priority.txt)
            __builtin_amdgcn_s_setprio(3);
#endif
            // ---- halo rows of the OTHER colour as of half-sweep j - 1 (tag epoch + j), from the neighbour strips
            bool bail = false;
#ifndef ISINGMC_STRIP_DEBUG_NOPOLL // (timing-only build, wrong results: what the waits cost)
            if (j > 0) {
                for (uint32_t i = tid; i < 2 * wpr; i += NT) {
                    const uint32_t side = i / wpr, w = i - side * wpr; // 0: my top halo = up_strip's bottom row; 1: my bottom halo
                    const strip_gu64 src = (strip_gu64)(rep_halo + strip_granule(g, side ? dn_strip : up_strip, 1 - colour, side ? 0 : 1) + w);
                    const uint32_t want = a.epoch + j;
                    const unsigned long long start = __builtin_amdgcn_s_memrealtime();
                    uint32_t spins = 0;
                    for (;;) {
                        const unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (uint32_t(v >> 32) == want) {
                            lds[(1 - colour) * PL + (side ? (S + 1) * wpr : 0) + w] = uint32_t(v);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                        if ((++spins & 63u) == 0 &&
                            (__builtin_amdgcn_s_memrealtime() - start > STRIP_TIMEOUT_TICKS ||
                             __hip_atomic_load((strip_gu32)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                            atomicOr(err, STRIP_ERR_TIMEOUT); // the grid was not resident at once (or a neighbour gave up): leave
                            bail = true;
                            break;
                        }
                    }
                }
            }
#endif
            if constexpr (NW > 1) {
                if (bail) red[8] = 1;
                __syncthreads(); // halo rows in place; everybody has finished the previous half-sweep's LDS stores
                if (red[8]) return;
            } else {
                if (__any(bail)) return; // a wave's LDS accesses execute in order: no barrier
            }
            bool odd = (y_global + colour) & 1u;
            if constexpr (NW > 1) odd = __builtin_amdgcn_readfirstlane(uint32_t(odd)) != 0; // wave-uniform: scalar branch
            uint32_t nw[4];
            strip_update_quad<PMJ>(lds, PL, g, colour, yl, col, Q, odd, t0 + k, key, vk, thr, PMJ ? jneg + size_t(colour) * 4 * g.wpp : nullptr,
                                   jneg_uniform, R, nw, measure && colour == 1, sat, up);
            // ---- publish my boundary rows of this colour (not after the last half-sweep: nobody waits for it)
            if ((top_row || bottom_row) && j + 1 < 2 * timesteps) {
                const unsigned long long tag = (unsigned long long)(a.epoch + j + 1) << 32;
                if (top_row) {
                    const strip_gu64 dst = (strip_gu64)(rep_halo + strip_granule(g, strip, colour, 0) + 4 * col);
#pragma unroll
                    for (int q = 0; q < 4; q++) __hip_atomic_store(dst + q, tag | nw[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (bottom_row) {
                    const strip_gu64 dst = (strip_gu64)(rep_halo + strip_granule(g, strip, colour, 1) + 4 * col);
#pragma unroll
                    for (int q = 0; q < 4; q++) __hip_atomic_store(dst + q, tag | nw[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#ifndef ISINGMC_STRIP_NO_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
#ifndef ISINGMC_STRIP_NO_PRECOMPUTE
            // the next half-sweep's words; in front of an exchange round they are drawn AFTER the replica's count has been
            // posted (the other strips and the partner replica wait for that post) and before the wait for the mailboxes
            if (j + 1 < 2 * timesteps && !(exchange && colour == 1)) quad_random(R, Q, 1 - colour, t0 + k + colour, key, vk);
#endif
        }
        if (measure) { // get_energy after this timestep (lattice.rs:454): the strips of a replica add up
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sat += __shfl_xor(sat, off);
            (void)up; // no up-spin count on lattices without a field (quad_measure): the second counter of a step stays zero
            unsigned long long s4 = sat;
            if constexpr (NW > 1) {
                __syncthreads(); // red[] of the previous timestep has been read
                if (lane == 0) red[wave] = sat;
                __syncthreads();
                s4 = (unsigned long long)red[0] + red[1] + red[2] + red[3];
            }
            if (tid == 0) {
                if (steps_out) atomicAdd(steps_out + (size_t(k) * n_replicas + r) * 2, s4);
                if (exchange) { // the replica's total for this round: the last strip to arrive posts it at the replica's rung
                    const unsigned long long round = lad.round0 + (k + 1) / lad.swap_every - 1;
                    unsigned long long *cnt = lad.round_counts + size_t(round & 1) * n_replicas + r;
                    const unsigned long long mine_add = s4 | (1ull << STRIP_ARRIVAL_SHIFT);
                    const unsigned long long old = atomicAdd(cnt, mine_add);
                    if ((old >> STRIP_ARRIVAL_SHIFT) + 1 == a.n_strips) {
                        const unsigned long long total = (old + mine_add) & ((1ull << STRIP_ARRIVAL_SHIFT) - 1);
                        // The counter of this parity is next added to two rounds later, by strips that have read this round's post
                        // first: the reset is ordered before the post (release) and the readers' later adds after their read of the
                        // post (acquire fence in the exchange below) -- once per replica and round, so the ordering costs nothing
                        __hip_atomic_store((strip_gu64)cnt, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store((strip_gu64)(lad.mail + size_t(round & 3) * lad.n_rungs + my_rung),
                                           ((unsigned long long)(uint32_t(round) + 1u) << 32) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (fin.counts && last_step) {
                    // ONE atomic carries the count and the arrival: whoever sees n_strips - 1 earlier arrivals holds the total
                    const unsigned long long mine_add = s4 | (1ull << STRIP_ARRIVAL_SHIFT);
                    const unsigned long long old = atomicAdd(fin.counts + r, mine_add);
                    if ((old >> STRIP_ARRIVAL_SHIFT) + 1 == a.n_strips) {
                        const long long total = (long long)((old + mine_add) & ((1ull << STRIP_ARRIVAL_SHIFT) - 1));
                        fin.energy_out[r] = fin.jabs * double(fin.n_bonds - 2 * total);
                        __hip_atomic_store((strip_gu64)(fin.counts + r), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        if (exchange) { // ---- one exchange round, decided by every strip of both partners alike (host twin: pt_swap_round)
#ifndef ISINGMC_STRIP_NO_PRECOMPUTE
            quad_random(R, Q, 0, t0 + k + 1, key, vk);
#endif
            const unsigned long long round = lad.round0 + (k + 1) / lad.swap_every - 1;
            const uint32_t parity = uint32_t(round & 1), want = uint32_t(round) + 1u;
            const unsigned long long *box = lad.mail + size_t(round & 3) * lad.n_rungs;
            uint32_t partner = 0xFFFFFFFFu; // pairs (i, i + 1) with i of the round's parity
            if ((my_rung & 1u) == parity) { if (my_rung + 1 < lad.n_rungs) partner = my_rung + 1; }
            else if (my_rung >= 1) partner = my_rung - 1;
            uint32_t new_rung = my_rung;
            bool bail = false;
            // lanes 0 and 1 wait for the two mailboxes side by side (the own one is also the round's barrier of this replica's strips)
            uint32_t got = 0;
            if (tid == 0) bail = !strip_wait_granule(box + my_rung, want, got, err);
            if (tid == 1 && partner != 0xFFFFFFFFu) bail = !strip_wait_granule(box + partner, want, got, err);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // pairs with the release of the posts read above
            const uint32_t sat_mine = __shfl(got, 0), sat_other = __shfl(got, 1);
            bail = __any(bail) != 0; // (wave 0; the other waves learn it through red[8])
            if (tid == 0) {
                if (!bail && partner != 0xFFFFFFFFu) {
                    const uint32_t lo = my_rung < partner ? my_rung : partner;
                    const double e_mine = lad.jabs * double(lad.n_bonds - 2 * (long long)sat_mine);
                    const double e_other = lad.jabs * double(lad.n_bonds - 2 * (long long)sat_other);
                    const double e_lo = my_rung < partner ? e_mine : e_other, e_hi = my_rung < partner ? e_other : e_mine;
                    const double d = (lad.ladder[lo] - lad.ladder[lo + 1]) * (e_lo - e_hi);
                    bool accept = d >= 0.0;
                    if (!accept) {
                        const uint4 rnd = philox4x32_10(make_uint4(lo, uint32_t(round), uint32_t(round >> 32), 0x50545357u),
                                                        make_uint2(lad.seed_lo, lad.seed_hi));
                        const unsigned long long x = ((unsigned long long)rnd.y << 32) | rnd.x;
                        accept = double(x >> 11) * (1.0 / 9007199254740992.0) < strip_det_exp(d);
                    }
                    if (accept) {
                        new_rung = partner;
                        if (strip == 0 && my_rung < partner) atomicAdd(lad.counters + 1, 1ull); // once per accepted pair
                    }
                }
            }
            if constexpr (NW > 1) {
                if (tid == 0) { red[9] = new_rung; if (bail) red[8] = 1; }
                __syncthreads();
                if (red[8]) return;
                my_rung = __builtin_amdgcn_readfirstlane(red[9]);
                __syncthreads(); // red[9] is free again
            } else {
                if (__builtin_amdgcn_readfirstlane(uint32_t(bail))) return;
                my_rung = __builtin_amdgcn_readfirstlane(new_rung);
            }
        }
    }
    if (LAD && strip == 0 && tid == 0) { // the ladder as this launch leaves it
        lad.perm_out[my_rung] = r;
        if (r == 0) lad.counters[0] = lad.round0 + (timesteps - 1) / lad.swap_every;
    }
    if constexpr (NW > 1) __syncthreads();
#pragma unroll
    for (uint32_t p = 0; p < 2; p++)
        reinterpret_cast<uint4 *>(mine + size_t(p) * g.wpp + size_t(y0) * wpr)[tid] = reinterpret_cast<const uint4 *>(lds + p * PL + wpr)[tid];
}

} // namespace isingmc
