// Replica-packed path for arbitrary graphs with uniform |J|, no fields and degree <= 6 (3-d cubic,
// triangular, honeycomb, diluted or odd-sized square lattices, random regular graphs ...): the general
// edge-list path of BASELINE config c5 at bit-sliced speed.  DESIGN.md S6.
//
// Layout: the SAME colour-major positions as the thread-per-site general path, but one 32-bit word per
// POSITION holding the spins of the 32 replicas of a group (bit b = replica 32g+b).  A neighbour gather
// is then one word for 32 replicas, the satisfied-bond count of a site is a bit-sliced 3-bit counter,
// and the acceptance test is the lattice kernel's: bit-planes of uniform prefixes compared MSB-first
// against per-class thresholds, ties resolved with 32 more bits.  Flipping a spin with k satisfied
// bonds out of deg costs dE = 2|J| m, m = 2k - deg: m <= 0 always flips, m = 1..6 flips with
// probability exp(-beta 2|J| m).  One thread owns a position-quad = the 4 positions p, p+64, p+128, p+192
// of a 256-position block (p = the quad's leader): Philox call pl yields plane pl for those 4 words, and
// for a fixed word q the 64 lanes of a wavefront touch 64 CONSECUTIVE positions -- every load (own word,
// ELL neighbour slot, neighbour gather on regular lattices) is a fully coalesced 256-byte access.
// Neighbours are stored ELL-style, slot-major: nbr_ell[i * n_pos + p] = position | (J>0) << 31, or
// PK_NO_NBR for the unused slots of a site with fewer than PK_MAX_DEG neighbours.
//
// Threshold table per replica group (uint32[PK_TAB_WORDS]), built on the host:
//   all[m-1]            bit r: replica r accepts class m outright (T = 2^THR_BITS)
//   tbw[m-1][p]         bit r: bit p (MSB first) of the top N_PLANES bits of T_m(beta_r)
//   lo[m-1][r]          low 32 bits of T_m(beta_r)
#pragma once
#include "lattice_kernels.hpp"
#include "packed_types.hpp"

namespace isingmc {

// satisfied-bond count of the 32 replicas at one position, bit-sliced (c0 = LSB), from its ELL slots x[] and the
// gathered neighbour words n[].  Fixed trip count with predication.
__device__ __forceinline__ void pk_count(const uint32_t x[PK_MAX_DEG], const uint32_t n[PK_MAX_DEG], uint32_t s,
                                         uint32_t &deg, uint32_t &c0, uint32_t &c1, uint32_t &c2)
{
    c0 = c1 = c2 = deg = 0;
#pragma unroll
    for (int i = 0; i < PK_MAX_DEG; i++) {
        const bool used = x[i] != PK_NO_NBR;
        deg += used;
        const uint32_t sat = used ? (s ^ n[i]) ^ ((x[i] >> 31) ? 0u : 0xFFFFFFFFu) : 0u; // J>0: satisfied when the spins differ
        const uint32_t k0 = c0 & sat;
        c0 ^= sat;
        const uint32_t k1 = c1 & k0;
        c1 ^= k0;
        c2 ^= k1;
    }
}

// replicas whose count equals k
__device__ __forceinline__ uint32_t pk_match(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t k)
{
    return ~((c0 ^ (0u - (k & 1u))) | (c1 ^ (0u - ((k >> 1) & 1u))) | (c2 ^ (0u - ((k >> 2) & 1u))));
}

// one colour class of one timestep; blockIdx.y = replica group
__attribute__((unused)) static __global__ __launch_bounds__(256, 8) void pk_sweep_kernel(uint32_t *__restrict__ state, const PkGraphDev G,
                                                       const uint32_t class_begin, const uint32_t class_end,
                                                       const uint64_t t, const uint2 *__restrict__ group_keys,
                                                       const uint32_t *__restrict__ tabs, const uint32_t tab_stride)
{
    const uint32_t g = blockIdx.y;
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x; // wave w of the class owns positions [256w, 256w+256)
    const uint32_t p0 = class_begin + 256 * (tid >> 6) + (tid & 63u); // the quad's leader
    if (p0 >= class_end) return;
    uint32_t *st = state + size_t(g) * G.n_pos;
    const uint32_t *tab = tabs + size_t(g) * tab_stride;
    const uint2 key = group_keys[g];
    const uint32_t PQ = p0; // Philox counter word of the quad

    // Memory phase, batched: the 4 own words and the 24 ELL slots of the quad go out together, then the 24
    // neighbour gathers -- two round trips per thread (word by word it was eight, and the kernel ran at a third
    // of its vector-ALU bound).  Buffer loads: descriptors in SGPRs, 32-bit offsets, data lands in the
    // register that held the offset.
    const __amdgpu_buffer_rsrc_t st_rsrc = __builtin_amdgcn_make_buffer_rsrc(st, 0, int(G.n_pos * sizeof(uint32_t)), 0x00020000);
    const __amdgpu_buffer_rsrc_t ell_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(G.nbr_ell), 0, int(uint32_t(PK_MAX_DEG) * G.n_pos * uint32_t(sizeof(uint32_t))), 0x00020000);
    // All 24 block headers first (scalar loads, one wait); then straight-line code: a slot is (own position + the header's
    // shift) | sign for translation blocks, all ones (PK_NO_NBR) for unused ones -- both from scalars -- and where the block
    // is neither, a branch holding nothing but a load overwrites it with the table entry (a header load or a use of the
    // loaded value per slot made the wave wait for memory 24 times).
    uint32_t own[4], x[4][PK_MAX_DEG], nb[4][PK_MAX_DEG];
    uint2 h[4][PK_MAX_DEG];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint2 *hdr = G.ell_hdr + size_t(__builtin_amdgcn_readfirstlane((p0 + 64 * q) >> 6)) * PK_MAX_DEG; // wave-uniform
#pragma unroll
        for (int i = 0; i < PK_MAX_DEG; i++) h[q][i] = hdr[i];
    }
#pragma unroll
    for (int q = 0; q < 4; q++) own[q] = __builtin_amdgcn_raw_buffer_load_b32(st_rsrc, 4 * (p0 + 64 * q), 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < PK_MAX_DEG; i++) {
            const uint32_t hx = __builtin_amdgcn_readfirstlane(h[q][i].x), hy = __builtin_amdgcn_readfirstlane(h[q][i].y);
            const uint32_t fill = (hx & 3u) == PK_HDR_UNUSED ? PK_NO_NBR : (hx & 0x80000000u); // scalar
            x[q][i] = (p0 + 64 * q + hy) | fill;
        }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < PK_MAX_DEG; i++)
            if ((__builtin_amdgcn_readfirstlane(h[q][i].x) & 3u) == PK_HDR_MIXED)
                x[q][i] = __builtin_amdgcn_raw_buffer_load_b32(ell_rsrc, 4 * (uint32_t(i) * G.n_pos + p0 + 64 * q), 0, 0);
#pragma unroll
    for (int i = 0; i < PK_MAX_DEG; i++)
#pragma unroll
        for (int q = 0; q < 4; q++)
            nb[q][i] = __builtin_amdgcn_raw_buffer_load_b32(
                st_rsrc, 4 * (x[q][i] == PK_NO_NBR ? p0 + 64 * q : (x[q][i] & 0x7FFFFFFFu)), 0, 0);

    // the (up to) three costly classes of a site: m_j = 2j + 2 - (deg & 1), k_j = deg/2 + 1 + j, i.e. table
    // row 2j for odd degrees and 2j+1 for even ones.  The table is uniform per workgroup (scalar loads);
    // the per-lane part is only the parity select.  Padding positions have degree 0 and flip freely:
    // nothing reads them.
    uint32_t eq[4][3], odd[4], lt[4], und[4], always[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t deg, c0, c1, c2;
        pk_count(x[q], nb[q], own[q], deg, c0, c1, c2);
        odd[q] = 0u - (deg & 1u);
        uint32_t costly = 0;
        lt[q] = 0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const uint32_t k = (deg >> 1) + 1 + j;
            eq[q][j] = k <= deg ? pk_match(c0, c1, c2, k) : 0u;
            costly |= eq[q][j];
            lt[q] |= eq[q][j] & ((odd[q] & tab[PK_TAB_ALL + 2 * j]) | (~odd[q] & tab[PK_TAB_ALL + 2 * j + 1]));
        }
        always[q] = ~costly; // m <= 0 flips outright
        und[q] = costly & ~lt[q];
    }

    const uint32_t c0 = uint32_t(t), c1 = PQ;
#pragma unroll
    for (int pl = 0; pl < N_PLANES; pl++) {
        const uint4 rnd = philox4x32_10(make_uint4(c0, c1, DOM_PK_SWEEP, ctr2(t, 0, pl)), key);
        const uint32_t rr[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
        uint32_t t_odd[3], t_even[3]; // wave-uniform threshold bit-planes of the six classes
#pragma unroll
        for (int j = 0; j < 3; j++) {
            t_odd[j] = tab[PK_TAB_TBW + (2 * j) * N_PLANES + pl];
            t_even[j] = tab[PK_TAB_TBW + (2 * j + 1) * N_PLANES + pl];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t tb = 0;
#pragma unroll
            for (int j = 0; j < 3; j++) tb |= eq[q][j] & ((odd[q] & t_odd[j]) | (~odd[q] & t_even[j]));
            const uint32_t decided = und[q] & (rr[q] ^ tb);
            lt[q] |= decided & tb;
            und[q] ^= decided;
        }
    }

    uint32_t acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = always[q] | lt[q];
#ifdef ISINGMC_TIMING_ONLY_NO_TIES // diagnostic build: what the tie stage costs (results are wrong without it)
    if (false) {
#else
    if (und[0] | und[1] | und[2] | und[3]) { // ties: n-th of the position-quad takes word n%4 of call N_PLANES + n/4
#endif
        uint32_t nres = 0;
        uint4 rnd = philox4x32_10(make_uint4(c0, c1, DOM_PK_SWEEP, ctr2(t, 0, N_PLANES)), key);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t mm = und[q];
            while (mm) {
                const uint32_t b = __ffs(mm) - 1;
                mm &= mm - 1;
                if (nres != 0 && (nres & 3u) == 0)
                    rnd = philox4x32_10(make_uint4(c0, c1, DOM_PK_SWEEP, ctr2(t, 0, N_PLANES + (nres >> 2))), key);
                const uint32_t j = ((eq[q][0] >> b) & 1u) ? 0u : ((eq[q][1] >> b) & 1u) ? 1u : 2u;
                const uint32_t row = 2 * j + 1 - (odd[q] & 1u);
                if (sel4(rnd, nres & 3u) < tab[PK_TAB_LO + row * 32 + b]) acc[q] |= 1u << b;
                nres++;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) __builtin_amdgcn_raw_buffer_store_b32(own[q] ^ acc[q], st_rsrc, 4 * (p0 + 64 * q), 0, 0);
}

// random start: position p is word q = (p & 255) >> 6 of its quad (leader p - 64 q):
// word = Philox(group key, (0, leader, 0, "PKIN"))[q]; padding positions 0
__attribute__((unused)) static __global__ __launch_bounds__(256) void pk_init_kernel(uint32_t *__restrict__ state, const PkGraphDev G,
                                                      const uint2 *__restrict__ group_keys, const uint32_t first_group)
{
    const uint32_t g = first_group + blockIdx.y;
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= G.n_pos) return;
    const uint32_t q = (p & 255u) >> 6;
    const uint4 rnd = philox4x32_10(make_uint4(0, p - 64 * q, 0, DOM_PK_INIT), group_keys[g]);
    state[size_t(g) * G.n_pos + p] = G.site[p] != PAD_SITE ? sel4(rnd, q) : 0u;
}

// the random start of ONE replica bit of a group (real-coupling path: a replica appended to a partly used group starts from
// its random start -- the bits a container does not own are not simulated there, see rj_sweep_kernel PARTIAL)
__attribute__((unused)) static __global__ __launch_bounds__(256) void pk_init_replica_kernel(uint32_t *__restrict__ state, const PkGraphDev G,
                                                              const uint2 *__restrict__ group_keys, const uint32_t g, const uint32_t bit)
{
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= G.n_pos) return;
    const uint32_t q = (p & 255u) >> 6;
    const uint4 rnd = philox4x32_10(make_uint4(0, p - 64 * q, 0, DOM_PK_INIT), group_keys[g]);
    const uint32_t v = G.site[p] != PAD_SITE ? (sel4(rnd, q) >> bit) & 1u : 0u;
    uint32_t *w = state + size_t(g) * G.n_pos + p;
    *w = (*w & ~(1u << bit)) | (v << bit);
}

// Directed satisfied-bond total and up-spin count per replica:  out[2r] += satisfied (directed),
// out[2r+1] += up spins.  Thread = position (stride 256 inside a chunk of 8192 positions), all 32 replicas of the
// group at once: the satisfied bonds of a position are counted bit-sliced (as in the sweep), the counts of 32
// positions are added into bit-sliced accumulators per thread (8 + 6 planes), and only then transposed:
// for every plane and replica bit one ballot + scalar popcount over the wavefront (the scalar unit is idle
// anyway).  ~1 vector instruction per position and 32 replicas; the previous version spent a lane per
// (position, replica) pair and took 7.3 ms for 256^3 x 64 replicas, 20x a sweep.
constexpr uint32_t PK_MEASURE_POS_PER_THREAD = 32; // 6 x 32 = 192 < 2^8 satisfied bonds, 32 < 2^6 up spins per thread
constexpr uint32_t PK_MEASURE_CHUNK = 256 * PK_MEASURE_POS_PER_THREAD;

// bit-sliced add of the K2-plane number x into the K-plane accumulator S (per bit lane), K2 <= K
template <int K, int K2>
__device__ __forceinline__ void bs_add(uint32_t (&S)[K], const uint32_t (&x)[K2])
{
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < K; i++) {
        const uint32_t a = S[i], b = i < K2 ? x[i] : 0u;
        S[i] = a ^ b ^ carry;
        carry = (a & b) | (carry & (a ^ b));
    }
}

// pos_per_thread <= PK_MEASURE_POS_PER_THREAD: fewer positions per thread give a mid-size graph enough workgroups
// (a thread walks its positions one after the other, each a chain of dependent loads)
// sat_end / sat_scale: on a 2-coloured (bipartite) graph every bond joins class 0 to class 1, so the bonds are counted from
// the class-0 positions only (sat_end = the end of class 0) and doubled (sat_scale = 2) into the same directed total -- half
// the neighbour gathers; otherwise sat_end = n_pos, sat_scale = 1.
__attribute__((unused)) static __global__ __launch_bounds__(256) void pk_measure_kernel(const uint32_t *__restrict__ state, const PkGraphDev G,
                                                         unsigned long long *__restrict__ out, const uint32_t n_replicas,
                                                         const uint32_t pos_per_thread, const uint32_t sat_end, const uint32_t sat_scale)
{
    __shared__ uint32_t red[2][4][32];
    const uint32_t g = blockIdx.y, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t *st = state + size_t(g) * G.n_pos;
    const __amdgpu_buffer_rsrc_t ell_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(G.nbr_ell), 0, int(uint32_t(PK_MAX_DEG) * G.n_pos * uint32_t(sizeof(uint32_t))), 0x00020000);
    uint32_t tot_sat = 0, tot_up = 0; // lane r < 32: running totals of replica r over this wave's positions
    const uint32_t chunk = 256 * pos_per_thread;
    for (uint32_t base = blockIdx.x * chunk; base < G.n_pos; base += gridDim.x * chunk) {
        uint32_t S[8] = {0, 0, 0, 0, 0, 0, 0, 0}, U[6] = {0, 0, 0, 0, 0, 0};
        for (uint32_t j = 0; j < pos_per_thread; j++) {
            const uint32_t p = base + 256 * j + threadIdx.x;
            if (p >= G.n_pos) break;
            if (G.site[p] == PAD_SITE) continue;
            const uint32_t s = st[p];
            const uint32_t up[1] = {s};
            bs_add(U, up);
            if (p >= sat_end) continue; // wave-uniform: class boundaries are multiples of 256
            uint32_t x[PK_MAX_DEG], n[PK_MAX_DEG];
            // (the lanes that left the loop above hold no header: the block index is the same for all lanes that remain)
            const uint2 *hdr = G.ell_hdr + size_t(__builtin_amdgcn_readfirstlane(p >> 6)) * PK_MAX_DEG;
            uint2 h[PK_MAX_DEG]; // the six headers first, then straight-line code with load-only branches (as pk_sweep_kernel)
#pragma unroll
            for (int i = 0; i < PK_MAX_DEG; i++) h[i] = hdr[i];
#pragma unroll
            for (int i = 0; i < PK_MAX_DEG; i++) {
                const uint32_t hx = __builtin_amdgcn_readfirstlane(h[i].x), hy = __builtin_amdgcn_readfirstlane(h[i].y);
                x[i] = (p + hy) | ((hx & 3u) == PK_HDR_UNUSED ? PK_NO_NBR : (hx & 0x80000000u));
            }
#pragma unroll
            for (int i = 0; i < PK_MAX_DEG; i++)
                if ((__builtin_amdgcn_readfirstlane(h[i].x) & 3u) == PK_HDR_MIXED)
                    x[i] = __builtin_amdgcn_raw_buffer_load_b32(ell_rsrc, 4 * (uint32_t(i) * G.n_pos + p), 0, 0);
#pragma unroll
            for (int i = 0; i < PK_MAX_DEG; i++) n[i] = st[x[i] == PK_NO_NBR ? p : (x[i] & 0x7FFFFFFFu)];
            uint32_t deg, c[3];
            pk_count(x, n, s, deg, c[0], c[1], c[2]);
            bs_add(S, c);
        }
        // transpose: replica r's count = sum over planes i of 2^i x (lanes of this wave with bit r of plane i set)
        for (uint32_t r = 0; r < 32; r++) {
            uint32_t sat_r = 0, up_r = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) sat_r += uint32_t(__popcll(__ballot((S[i] >> r) & 1u))) << i;
#pragma unroll
            for (int i = 0; i < 6; i++) up_r += uint32_t(__popcll(__ballot((U[i] >> r) & 1u))) << i;
            if (lane == r) { tot_sat += sat_r; tot_up += up_r; }
        }
    }
    if (lane < 32) { red[0][wave][lane] = tot_sat; red[1][wave][lane] = tot_up; }
    __syncthreads();
    if (threadIdx.x < 32) {
        const uint32_t r = 32 * g + threadIdx.x;
        unsigned long long s = 0, u = 0;
        for (int k = 0; k < 4; k++) { s += red[0][k][threadIdx.x]; u += red[1][k][threadIdx.x]; }
        if (r < n_replicas && (s | u)) {
            atomicAdd(out + 2 * size_t(r), s * sat_scale);
            atomicAdd(out + 2 * size_t(r) + 1, u);
        }
    }
}

// write one replica's spins (bits[] packed by position, 32 positions per word) into bit `bit` of group g
__attribute__((unused)) static __global__ __launch_bounds__(256) void pk_set_replica_kernel(uint32_t *__restrict__ state, const uint32_t n_pos,
                                                             const uint32_t *__restrict__ bits, const uint32_t g,
                                                             const uint32_t bit)
{
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pos) return;
    uint32_t *w = state + size_t(g) * n_pos + p;
    const uint32_t v = (bits[p >> 5] >> (p & 31u)) & 1u;
    *w = (*w & ~(1u << bit)) | (v << bit);
}

// ---- tempering on the stream (isingmc_pt_* on a packed container) -------------------------------------------------------
// Energies of the local slots from the measurement counters: E = |J| (undirected bonds - directed satisfied count) + self loops --
// the arithmetic of the host's pk_energy, hence the same bits.
__attribute__((unused)) static __global__ void pk_energy_from_counts_kernel(const unsigned long long *__restrict__ meas, const uint32_t first_slot,
                                                                            const uint32_t n, const double jabs, const double half_directed,
                                                                            const double self_energy, double *__restrict__ out)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) out[r] = jabs * (half_directed - double((long long)meas[2 * size_t(first_slot + r)])) + self_energy;
}

// Threshold tables of every group from per-slot thresholds (slot_thr[slot][m - 1] = T_m(beta of the slot), written by the exchange
// kernel): what pk_fill_table builds on the host for isingmc_states_set_betas, bit for bit.  One wavefront per group; lane r < 32 =
// replica bit r; slots at or beyond n_owned take the last owned slot's thresholds (as the host does).
__attribute__((unused)) static __global__ void pk_tables_from_slots_kernel(const unsigned long long *__restrict__ slot_thr, const uint32_t n_owned,
                                                                           uint32_t *__restrict__ tabs)
{
    const uint32_t g = blockIdx.x, lane = threadIdx.x; // 64 threads
    uint32_t *tab = tabs + size_t(g) * PK_TAB_WORDS;
    const uint32_t slot = min(32u * g + (lane & 31u), n_owned - 1u);
    for (uint32_t m = 0; m < uint32_t(PK_MAX_DEG); m++) {
        const unsigned long long T = lane < 32 ? slot_thr[size_t(slot) * PK_MAX_DEG + m] : 0ull;
        const uint32_t all = uint32_t(__ballot(lane < 32 && (T >> THR_BITS) != 0));
        if (lane == 0) tab[PK_TAB_ALL + m] = all;
        const uint32_t hi = uint32_t(T >> 32) & ((1u << N_PLANES) - 1);
        for (int p = 0; p < N_PLANES; p++) {
            const uint32_t w = uint32_t(__ballot(lane < 32 && ((hi >> (N_PLANES - 1 - p)) & 1u)));
            if (lane == 0) tab[PK_TAB_TBW + m * N_PLANES + p] = w;
        }
        if (lane < 32) tab[PK_TAB_LO + m * 32 + lane] = uint32_t(T);
    }
}

} // namespace isingmc
