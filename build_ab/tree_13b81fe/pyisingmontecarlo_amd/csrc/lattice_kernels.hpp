// Checkerboard Metropolis kernels for a periodic W x H square lattice with uniform |J|
// (replaces the serial do_time_step loop of lattice.rs:204-207 for recognised lattices).
//
// Layout: colour c = (x+y)&1; the colour-c sites of row y are x = 2i + ((y+c)&1);
// plane c stores row y as wpr = W/64 words, spin i at bit (i&31) of word y*wpr + (i>>5);
// a replica = plane 0 followed by plane 1.  One thread owns one quad = 4 consecutive words
// (128 spins) of the plane being updated: one 16-byte load / store per operand.
//
// Acceptance: a spin with k satisfied bonds flips always for k <= 2 and with
// probability exp(-beta 2|J|(2k-4)) for k = 3, 4.  All 128 decisions of a quad are taken
// bit-sliced: Philox call p (p = 0..N_PLANES-1) yields bit-plane p of the spins' uniform prefixes,
// compared MSB-first against the top N_PLANES bits of the fixed-point threshold of each spin's class.
// The few spins whose prefix ties the threshold (2^-N_PLANES of them) are resolved with one 32-bit
// Philox word each.  Everything is integer: the CPU oracle reproduces the configurations bit for bit.
#pragma once
#include "philox.hpp"
#include <type_traits>

namespace isingmc {

struct LatGeom {
    uint32_t W, H;
    uint32_t wpr;    // words per colour-row = W / 64
    uint32_t wpp;    // words per plane = H * wpr
    uint32_t nquads; // wpp / 4
    int32_t cols_log2; // log2(quads per row) when the division-free, parity-uniform thread mapping applies, else -1
};

struct LatThr {
    uint64_t T3, T4; // floor(exp(-beta dE) 2^THR_BITS) for k = 3, 4; 2^THR_BITS = always accept
};

#ifndef ISINGMC_N_PLANES
#define ISINGMC_N_PLANES 7
#endif
constexpr int N_PLANES = ISINGMC_N_PLANES;
constexpr int THR_BITS = N_PLANES + 32; // acceptance probabilities are fixed-point with this many bits

__device__ __forceinline__ uint32_t sel4(uint4 v, uint32_t i)
{
    return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

// neighbour words of one quad: up / down / centre / side, as the kernel consumes them
struct QuadNbr {
    uint32_t up[4], dn[4], ce[4], si[4];
};

// Thread -> quad.  With 2^k quads per row the mapping needs no division and keeps the row parity
// (which decides whether the side neighbour is i+1 or i-1) uniform per wavefront: a pair of waves
// shares 2*rpw consecutive rows, wave 0 takes the even ones, wave 1 the odd ones (rpw = 64 >> k rows
// per wave; for k >= 6 a wave never leaves its row).  Any bijection is valid: the Philox counters are
// functions of the quad index, never of the thread index.
template <bool UNI>
__device__ __forceinline__ void thread_to_quad(const LatGeom &g, uint32_t gid, uint32_t &Q, uint32_t &y,
                                               uint32_t &xw)
{
    if constexpr (UNI) {
        const uint32_t cl = uint32_t(g.cols_log2), col = gid & ((1u << cl) - 1);
        if (cl >= 6) {
            y = gid >> cl;
        } else {
            const uint32_t wave = gid >> 6, j = (gid & 63u) >> cl;
            y = ((wave >> 1) << (7 - cl)) + 2 * j + (wave & 1u);
        }
        Q = (y << cl) + col;
        xw = 4 * col;
    } else {
        Q = gid;
        y = (4 * Q) / g.wpr;
        xw = 4 * Q - y * g.wpr;
    }
}

// Where the two planes of a replica live.  PtrPlanes: plain pointers (LDS in the resident kernel, or global).
// BufPlanes: one buffer descriptor per replica (wave-uniform, in SGPRs) + 32-bit byte offsets -- the address
// arithmetic of the streaming kernel shrinks from 64-bit VALU adds per access to one 32-bit offset.
struct PtrPlanes {
    uint32_t *own;
    const uint32_t *oth;
    __device__ __forceinline__ uint4 own4(uint32_t w) const { return *reinterpret_cast<const uint4 *>(own + w); }
    __device__ __forceinline__ uint4 oth4(uint32_t w) const { return *reinterpret_cast<const uint4 *>(oth + w); }
    __device__ __forceinline__ uint32_t own1(uint32_t w) const { return own[w]; }
    __device__ __forceinline__ uint32_t oth1(uint32_t w) const { return oth[w]; }
    __device__ __forceinline__ void store4(uint32_t w, uint4 v) const { *reinterpret_cast<uint4 *>(own + w) = v; }
    __device__ __forceinline__ void store1(uint32_t w, uint32_t v) const { own[w] = v; }
};

struct BufPlanes {
    __amdgpu_buffer_rsrc_t rsrc; // the replica's 2 * wpp words
    uint32_t own_off, oth_off;   // byte offsets of the two planes
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ uint4 ld4(uint32_t byte_off) const
    {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    __device__ __forceinline__ uint4 own4(uint32_t w) const { return ld4(own_off + 4 * w); }
    __device__ __forceinline__ uint4 oth4(uint32_t w) const { return ld4(oth_off + 4 * w); }
    __device__ __forceinline__ uint32_t own1(uint32_t w) const { return __builtin_amdgcn_raw_buffer_load_b32(rsrc, own_off + 4 * w, 0, 0); }
    __device__ __forceinline__ uint32_t oth1(uint32_t w) const { return __builtin_amdgcn_raw_buffer_load_b32(rsrc, oth_off + 4 * w, 0, 0); }
    __device__ __forceinline__ void store4(uint32_t w, uint4 v) const
    {
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rsrc, own_off + 4 * w, 0, 0);
    }
    __device__ __forceinline__ void store1(uint32_t w, uint32_t v) const { __builtin_amdgcn_raw_buffer_store_b32(v, rsrc, own_off + 4 * w, 0, 0); }
};

// side-neighbour words from the centre words and the one word beyond the quad (in n.si[0])
__device__ __forceinline__ void side_words(QuadNbr &n, bool odd)
{
    const uint32_t sw = n.si[0];
    if (odd) {
        n.si[0] = (n.ce[0] >> 1) | (n.ce[1] << 31);
        n.si[1] = (n.ce[1] >> 1) | (n.ce[2] << 31);
        n.si[2] = (n.ce[2] >> 1) | (n.ce[3] << 31);
        n.si[3] = (n.ce[3] >> 1) | (sw << 31);
    } else {
        n.si[3] = (n.ce[3] << 1) | (n.ce[2] >> 31);
        n.si[2] = (n.ce[2] << 1) | (n.ce[1] >> 31);
        n.si[1] = (n.ce[1] << 1) | (n.ce[0] >> 31);
        n.si[0] = (n.ce[0] << 1) | (sw >> 31);
    }
}

// Loads the 4 own words and the neighbour words of quad Q (row y, first word xw) in plane `colour`.
template <bool VEC, bool UNI, typename Mem>
__device__ __forceinline__ void load_quad(const Mem &mem, const LatGeom &g,
                                          uint32_t colour, uint32_t Q, uint32_t y, uint32_t xw, uint32_t own[4],
                                          QuadNbr &n, uint32_t widx[4])
{
    if constexpr (VEC) { // wpr % 4 == 0: the quad lies inside one row
        const uint32_t w0 = 4 * Q;
        const uint32_t yu = (y == 0 ? g.H : y) - 1, yd = (y + 1 == g.H) ? 0 : y + 1;
        // horizontal neighbours have compact indices {i, i+1} on odd rows of this colour, {i-1, i} on even
        // ones: wave-uniform under the 2^k mapping (scalar branch), per-lane otherwise
        bool odd = (y + colour) & 1u;
        if constexpr (UNI) odd = __builtin_amdgcn_readfirstlane(uint32_t(odd));
        const uint32_t sx = odd ? (xw + 4 == g.wpr ? 0 : xw + 4) : (xw == 0 ? g.wpr : xw) - 1;
        // row * wpr: a shift under the 2^k mapping (wpr = 4 << cols_log2), else a quarter-rate multiply
        const auto row_base = [&](uint32_t row) { return UNI ? row << (uint32_t(g.cols_log2) + 2) : row * g.wpr; };
        const uint4 o4 = mem.own4(w0);
        const uint4 c4 = mem.oth4(w0);
        const uint4 u4 = mem.oth4(row_base(yu) + xw);
        const uint4 d4 = mem.oth4(row_base(yd) + xw);
        const uint32_t sw = mem.oth1(row_base(y) + sx); // issued with the other loads, not behind a branch
        own[0] = o4.x; own[1] = o4.y; own[2] = o4.z; own[3] = o4.w;
        n.ce[0] = c4.x; n.ce[1] = c4.y; n.ce[2] = c4.z; n.ce[3] = c4.w;
        n.up[0] = u4.x; n.up[1] = u4.y; n.up[2] = u4.z; n.up[3] = u4.w;
        n.dn[0] = d4.x; n.dn[1] = d4.y; n.dn[2] = d4.z; n.dn[3] = d4.w;
        n.si[0] = sw;
        side_words(n, odd);
#pragma unroll
        for (int q = 0; q < 4; q++) widx[q] = w0 + q;
    } else { // narrow lattices (wpr = 1, 2, ...): the quad's words sit in different rows
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t w = 4 * Q + q;
            const uint32_t yy = w / g.wpr, xx = w - yy * g.wpr;
            const uint32_t yu = (yy == 0 ? g.H : yy) - 1, yd = (yy + 1 == g.H) ? 0 : yy + 1;
            const uint32_t row = yy * g.wpr;
            widx[q] = w;
            own[q] = mem.own1(w);
            n.ce[q] = mem.oth1(w);
            n.up[q] = mem.oth1(yu * g.wpr + xx);
            n.dn[q] = mem.oth1(yd * g.wpr + xx);
            if ((yy + colour) & 1) {
                const uint32_t nxt = mem.oth1(row + (xx + 1 == g.wpr ? 0 : xx + 1));
                n.si[q] = (n.ce[q] >> 1) | (nxt << 31);
            } else {
                const uint32_t prv = mem.oth1(row + (xx == 0 ? g.wpr : xx) - 1);
                n.si[q] = (n.ce[q] << 1) | (prv >> 31);
            }
        }
    }
}

// "bond satisfied" masks of the four bonds of each spin of word q.
// jneg planes hold 1 where J < 0 (ferromagnetic): satisfied = ~(s ^ n) ^ jpos = s ^ n ^ jneg.
template <bool PMJ>
__device__ __forceinline__ void bond_masks(const uint32_t own, const QuadNbr &n, int q,
                                           const uint32_t *__restrict__ jneg, uint32_t wpp,
                                           uint32_t widx, uint32_t jneg_uniform, uint32_t &a0,
                                           uint32_t &a1, uint32_t &a2, uint32_t &a3)
{
    if constexpr (PMJ) {
        a0 = own ^ n.up[q] ^ jneg[widx];
        a1 = own ^ n.dn[q] ^ jneg[wpp + widx];
        a2 = own ^ n.ce[q] ^ jneg[2 * wpp + widx];
        a3 = own ^ n.si[q] ^ jneg[3 * wpp + widx];
    } else {
        const uint32_t o = own ^ jneg_uniform;
        a0 = o ^ n.up[q];
        a1 = o ^ n.dn[q];
        a2 = o ^ n.ce[q];
        a3 = o ^ n.si[q];
    }
}

// Bit-sliced decision state of one quad: class masks (3 / 4 satisfied bonds), "prefix already smaller than
// the threshold" and "prefix still equal to it" per spin.
struct QuadState {
    uint32_t eq3[4], eq4[4], lt[4], und[4];
};

// the wave-uniform threshold data of a replica as the kernels consume it
struct ThrBits {
    uint32_t hi3, hi4; // top N_PLANES bits of T3, T4
    uint32_t lo3, lo4; // low 32 bits
    bool all3, all4;   // T = 2^THR_BITS: accepted outright
    bool any_all;
};

__device__ __forceinline__ ThrBits thr_bits(const LatThr thr)
{
    ThrBits b;
    // thresholds have THR_BITS = N_PLANES + 32 bits: the top N_PLANES are compared bit-sliced, the
    // low 32 against one residual Philox word
    // the high words go through readfirstlane first: thr is per replica (wave-uniform), and without it the
    // compiler fuses the tests below into 64-bit compares, which exist only on the vector unit
    const uint32_t h3 = __builtin_amdgcn_readfirstlane(uint32_t(thr.T3 >> 32));
    const uint32_t h4 = __builtin_amdgcn_readfirstlane(uint32_t(thr.T4 >> 32));
    b.all3 = (h3 >> N_PLANES) != 0;
    b.all4 = (h4 >> N_PLANES) != 0;
    b.any_all = ((h3 | h4) >> N_PLANES) != 0;
    b.hi3 = h3 & ((1u << N_PLANES) - 1);
    b.hi4 = h4 & ((1u << N_PLANES) - 1);
    b.lo3 = uint32_t(thr.T3);
    b.lo4 = uint32_t(thr.T4);
    return b;
}

// bit-sliced count of satisfied bonds -> class masks of the 128 spins of a quad
// +-J: the 16 sign words of quad Q (4 directions x 4 words), loaded in one batch BEFORE the spin words so
// that one wait covers both (behind a branch per word they were four dependent round trips to L2)
struct QuadSigns {
    uint32_t w[4][4]; // [word][direction: up, down, centre, side]
};

template <bool PMJ>
__device__ __forceinline__ void load_signs(const uint32_t *__restrict__ jn, const LatGeom &g, const uint32_t Q, QuadSigns &js)
{
    if constexpr (PMJ) {
#pragma unroll
        for (int d = 0; d < 4; d++) { // wpp is a multiple of 4 (fast-path condition): every quad is 16-byte aligned
            const uint4 v = *reinterpret_cast<const uint4 *>(jn + size_t(d) * g.wpp + 4 * size_t(Q));
            js.w[0][d] = v.x; js.w[1][d] = v.y; js.w[2][d] = v.z; js.w[3][d] = v.w;
        }
    }
}

template <bool PMJ>
__device__ __forceinline__ void quad_classes(const uint32_t own[4], const QuadNbr &n, const uint32_t widx[4], const LatGeom &g,
                                             const ThrBits &tb, const QuadSigns &js, const uint32_t jneg_uniform,
                                             QuadState &st)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t a0, a1, a2, a3;
        if constexpr (PMJ) { // jneg planes hold 1 where J < 0: satisfied = own ^ neighbour ^ jneg
            a0 = own[q] ^ n.up[q] ^ js.w[q][0];
            a1 = own[q] ^ n.dn[q] ^ js.w[q][1];
            a2 = own[q] ^ n.ce[q] ^ js.w[q][2];
            a3 = own[q] ^ n.si[q] ^ js.w[q][3];
        } else {
            bond_masks<false>(own[q], n, q, nullptr, g.wpp, widx[q], jneg_uniform, a0, a1, a2, a3);
        }
        const uint32_t s01 = a0 ^ a1, c01 = a0 & a1, s23 = a2 ^ a3, c23 = a2 & a3;
        st.eq4[q] = c01 & c23;
        st.eq3[q] = __builtin_amdgcn_bitop3_b32(c01, s23, c23 & s01, 0xEA); // (c01 & s23) | (c23 & s01)
        // pin the two class masks in registers: hipcc otherwise re-derives them from the bond masks inside
        // every plane (7 instead of 5 instructions per word and plane)
        asm("" : "+v"(st.eq3[q]), "+v"(st.eq4[q]));
    }
    if (tb.any_all) { // threshold 2^THR_BITS (beta = 0 ...): the class flips outright, like k <= 2; scalar branch, rarely taken
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (tb.all3) st.eq3[q] = 0;
            if (tb.all4) st.eq4[q] = 0;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        st.lt[q] = 0;
        st.und[q] = st.eq3[q] | st.eq4[q];
    }
}

// N_PLANES bit-planes of the uniform prefixes compared against the top N_PLANES threshold bits of each spin's
// class, for NQ quads at once: afterwards lt = "prefix < threshold bits", und = "prefix == threshold bits" (a tie).
// Philox counter = (t_lo, quad, domain, ctr2): the lane-varying quad index sits in a NON-multiplied word, which
// keeps round 1 of every call on the scalar unit and rounds 2-3 at one vector multiply each; the call index
// sits in the OTHER non-multiplied word, so those two vector multiplies do not depend on it and are shared by
// all calls of a quad (common subexpressions): 2 + 14 per call instead of 16.
// The comparison runs from the LEAST significant plane up: with r the random bit and tb the threshold bit of a
// spin, lt' = (~r & tb) | (~(r ^ tb) & lt) and eq' = eq & ~(r ^ tb) are both 3-input functions -- two
// v_bitop3_b32 per word and plane whatever the threshold bits (MSB first needs three, and a fourth register
// copy on the planes whose threshold bits are both 0).  Plane p is Philox call p either way: same decisions.
// NQ > 1 decides several quads together (the wave-uniform work is then issued once); measured, a two-quad
// kernel built on it ran exactly as fast as the one-quad kernel (the kernel is bound by VALU cycles, which
// are the same per quad, and the scalar unit runs beside it), so only NQ = 1 is instantiated.
// one plane of the comparison for the four words rr[] of a quad (plane p = Philox call p): what quad_planes does per
// plane, for callers that hold the random words already (kept apart from quad_planes: routing the streaming kernels
// through it cost lat_sweep_loop_kernel two registers, 66 instead of 64 = a wave per SIMD)
__device__ __forceinline__ void plane_step(QuadState &st, const uint32_t rr[4], const int p, const ThrBits &tb)
{
    // threshold bit of this plane for class 3 / class 4: wave-uniform, so the per-spin threshold word is
    // one of {0, eq3, eq4, eq3|eq4} -- scalar branches pick the register
    const auto step = [&](int q, uint32_t tbw) {
        st.lt[q] = __builtin_amdgcn_bitop3_b32(rr[q], tbw, st.lt[q], 0x8E);   // (~r & tb) | (~(r ^ tb) & lt)
        st.und[q] = __builtin_amdgcn_bitop3_b32(st.und[q], rr[q], tbw, 0x90); // eq & ~(r ^ tb)
    };
    // one scalar selector per plane (the scalar unit runs beside the vector ALU; as two nested bool tests the
    // compiler parked the second bool in a VGPR: a v_cndmask and a v_cmp per plane)
    const uint32_t sel = __builtin_amdgcn_readfirstlane(((tb.hi3 >> (N_PLANES - 1 - p)) & 1u) | (((tb.hi4 >> (N_PLANES - 1 - p)) & 1u) << 1));
    if (sel == 3) {
#pragma unroll
        for (int q = 0; q < 4; q++) step(q, st.eq3[q] | st.eq4[q]);
    } else if (sel == 1) {
#pragma unroll
        for (int q = 0; q < 4; q++) step(q, st.eq3[q]);
    } else if (sel == 2) {
#pragma unroll
        for (int q = 0; q < 4; q++) step(q, st.eq4[q]);
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            st.lt[q] &= ~rr[q];
            st.und[q] &= ~rr[q];
        }
    }
}

template <int NQ>
__device__ __forceinline__ void quad_planes(QuadState (&st)[NQ], const uint32_t (&Q)[NQ], const uint32_t colour, const uint64_t t,
                                            const uint2 key, const PhiloxVKeys &vk, const ThrBits &tb)
{
    const uint32_t c0 = uint32_t(t);
#pragma unroll
    for (int p = N_PLANES - 1; p >= 0; p--) {
        uint32_t rr[NQ][4];
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            const uint4 rnd = philox4x32_10(make_uint4(c0, Q[j], DOM_LAT_SWEEP, ctr2(t, colour, p)), key, vk);
            rr[j][0] = rnd.x; rr[j][1] = rnd.y; rr[j][2] = rnd.z; rr[j][3] = rnd.w;
        }
        // threshold bit of this plane for class 3 / class 4: wave-uniform, so the per-spin threshold word is
        // one of {0, eq3, eq4, eq3|eq4} -- scalar branches pick the register
        auto step = [&](int j, int q, uint32_t tbw) {
            st[j].lt[q] = __builtin_amdgcn_bitop3_b32(rr[j][q], tbw, st[j].lt[q], 0x8E);   // (~r & tb) | (~(r ^ tb) & lt)
            st[j].und[q] = __builtin_amdgcn_bitop3_b32(st[j].und[q], rr[j][q], tbw, 0x90); // eq & ~(r ^ tb)
        };
        // one scalar selector per plane (the scalar unit runs beside the vector ALU; as two nested bool tests the
        // compiler parked the second bool in a VGPR: a v_cndmask and a v_cmp per plane)
        const uint32_t sel = __builtin_amdgcn_readfirstlane(((tb.hi3 >> (N_PLANES - 1 - p)) & 1u) | (((tb.hi4 >> (N_PLANES - 1 - p)) & 1u) << 1));
        if (sel == 3) {
#pragma unroll
            for (int j = 0; j < NQ; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) step(j, q, st[j].eq3[q] | st[j].eq4[q]);
        } else if (sel == 1) {
#pragma unroll
            for (int j = 0; j < NQ; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) step(j, q, st[j].eq3[q]);
        } else if (sel == 2) {
#pragma unroll
            for (int j = 0; j < NQ; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) step(j, q, st[j].eq4[q]);
        } else {
#pragma unroll
            for (int j = 0; j < NQ; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    st[j].lt[q] &= ~rr[j][q];
                    st[j].und[q] &= ~rr[j][q];
                }
        }
    }
}

// The random words of a quad's half-sweep do not depend on the spins: a kernel that has to wait for its neighbours
// (strip_kernels.hpp) draws them BEFORE the wait.  rr[p] = plane p, tie = the first residual call.
// (Two calls written side by side, their rounds interleaved by volatile asm, ran 1-2 % SLOWER at 4 waves per SIMD:
// a lone wave is bound by its ~4-cycle issue interval, not by the multiply -> xor dependency chain.)
struct QuadRandom {
    uint32_t rr[N_PLANES][4];
    uint4 tie;
};

__device__ __forceinline__ void quad_random(QuadRandom &R, const uint32_t Q, const uint32_t colour, const uint64_t t, const uint2 key,
                                            const PhiloxVKeys &vk)
{
    const uint32_t c0 = uint32_t(t);
#pragma unroll
    for (int p = 0; p < N_PLANES; p++) {
        const uint4 rnd = philox4x32_10(make_uint4(c0, Q, DOM_LAT_SWEEP, ctr2(t, colour, p)), key, vk);
        R.rr[p][0] = rnd.x; R.rr[p][1] = rnd.y; R.rr[p][2] = rnd.z; R.rr[p][3] = rnd.w;
    }
    R.tie = philox4x32_10(make_uint4(c0, Q, DOM_LAT_SWEEP, ctr2(t, colour, N_PLANES)), key, vk);
}

// residual stage: spins whose prefix equals the threshold's top bits (ties) draw 32 more bits; acc = flips
__device__ __forceinline__ void quad_ties(const QuadState &st, const uint32_t Q, const uint32_t colour, const uint64_t t,
                                          const uint2 key, const PhiloxVKeys &vk, const ThrBits &tb, uint32_t acc[4],
                                          const uint4 *first_call = nullptr)
{
    const uint32_t c0 = uint32_t(t), c1 = Q;
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = __builtin_amdgcn_bitop3_b32(st.eq3[q], st.eq4[q], st.lt[q], 0xAB); // ~(eq3|eq4) | lt
#ifdef ISINGMC_TIMING_ONLY_NO_TIES // diagnostic build: what the tie stage costs (results are wrong without it)
    if (false) {
#else
    if (st.und[0] | st.und[1] | st.und[2] | st.und[3]) {
#endif
        // the first residual call is hoisted: inside the divergent per-word loops below it would be
        // issued once per loop (up to 4x per wave) instead of once
        const uint4 rnd = first_call ? *first_call : philox4x32_10(make_uint4(c0, c1, DOM_LAT_SWEEP, ctr2(t, colour, N_PLANES)), key, vk);
#ifdef ISINGMC_TIMING_ONLY_NO_TIE_LOOPS // diagnostic build (results are wrong): the residual call kept, the per-tie loops replaced
        // by ONE compare per word -- the floor of any reorganisation of the loops (wave-level compaction included)
        {
            const uint32_t rr4[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q] |= st.und[q] & (0u - uint32_t(rr4[q] < ((st.und[q] & st.eq4[q]) ? tb.lo4 : tb.lo3)));
            return;
        }
#endif
        const uint32_t n_ties = __popc(st.und[0]) + __popc(st.und[1]) + __popc(st.und[2]) + __popc(st.und[3]);
        if (n_ties <= 4) {
            // all but ~0.1 % of the quads: the ties consume the four words of this one call in order.  The words
            // rotate through r0 (three full-rate moves per tie) instead of being selected by a lane-varying
            // index, and no loop iteration tests for a refill
            uint32_t r0 = rnd.x, r1 = rnd.y, r2 = rnd.z, r3 = rnd.w;
            // the two residual thresholds in VGPRs, once: v_cndmask takes its mask from the constant bus, so
            // the compiler would otherwise re-materialise both scalars in front of every select
            uint32_t lo3v, lo4v;
            asm volatile("v_mov_b32 %0, %1" : "=v"(lo3v) : "s"(tb.lo3));
            asm volatile("v_mov_b32 %0, %1" : "=v"(lo4v) : "s"(tb.lo4));
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t m = st.und[q];
                while (m) {
                    const uint32_t bit = m & (0u - m);
                    m ^= bit;
                    const uint32_t lo = (bit & st.eq4[q]) ? lo4v : lo3v;
                    if (r0 < lo) acc[q] |= bit;
                    r0 = r1; r1 = r2; r2 = r3;
                }
            }
        } else { // 5 or more ties in one quad: further calls, word n%4 of call N_PLANES + n/4
            uint32_t nres = 0;
            uint4 w = rnd;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t m = st.und[q];
                while (m) {
                    const uint32_t b = __ffs(m) - 1;
                    m &= m - 1;
                    if (nres != 0 && (nres & 3u) == 0)
                        w = philox4x32_10(make_uint4(c0, c1, DOM_LAT_SWEEP, ctr2(t, colour, N_PLANES + (nres >> 2))), key, vk);
                    const uint32_t lo = ((st.eq4[q] >> b) & 1u) ? tb.lo4 : tb.lo3;
                    if (sel4(w, nres & 3u) < lo) acc[q] |= 1u << b;
                    nres++;
                }
            }
        }
    }
}

// The 128 flip decisions of quad Q from its loaded words: acc[q] bit b = 1 where the spin flips.
template <bool PMJ>
__device__ __forceinline__ void quad_flips(const uint32_t own[4], const QuadNbr &n, const uint32_t widx[4], const LatGeom &g,
                                           const uint32_t colour, const uint64_t t, const uint2 key, const PhiloxVKeys &vk,
                                           const LatThr thr, const QuadSigns &js, const uint32_t jneg_uniform,
                                           const uint32_t Q, uint32_t acc[4])
{
    const ThrBits tb = thr_bits(thr);
    QuadState st[1];
    const uint32_t Qs[1] = {Q};
    quad_classes<PMJ>(own, n, widx, g, tb, js, jneg_uniform, st[0]);
    quad_planes<1>(st, Qs, colour, t, key, vk, tb);
    quad_ties(st[0], Q, colour, t, key, vk, tb, acc);
}

// the same decisions from random words drawn earlier (quad_random)
template <bool PMJ>
__device__ __forceinline__ void quad_flips_pre(const uint32_t own[4], const QuadNbr &n, const uint32_t widx[4], const LatGeom &g,
                                               const uint32_t colour, const uint64_t t, const uint2 key, const PhiloxVKeys &vk,
                                               const LatThr thr, const QuadSigns &js, const uint32_t jneg_uniform, const uint32_t Q,
                                               const QuadRandom &R, uint32_t acc[4])
{
    const ThrBits tb = thr_bits(thr);
    QuadState st;
    quad_classes<PMJ>(own, n, widx, g, tb, js, jneg_uniform, st);
#pragma unroll
    for (int p = N_PLANES - 1; p >= 0; p--) plane_step(st, R.rr[p], p, tb);
    quad_ties(st, Q, colour, t, key, vk, tb, acc, &R.tie);
}

// One Metropolis update of the 128 spins of a quad (thread index gid -> quad via thread_to_quad) of the
// plane `own_plane`, reading its neighbours from `oth_plane`.  The planes may live in HBM (sweep kernel)
// or in LDS (resident kernel): the function only sees pointers.
// Streaming kernels under the 2^k mapping: thread -> quad and the five load offsets with as little VECTOR
// arithmetic as possible (the kernel is bound by vector-ALU cycles; the scalar unit is idle beside it).
// The wave index, the row parity and the "does this wave touch row 0 or row H-1" test are wave-uniform and
// live on the scalar unit; the plane offsets ride in the buffer instructions' scalar offset; rows wrap
// per lane only in the two waves per plane that contain a boundary row; the row length is a power of two, so
// the side word's wrap is an AND.  Same quads, same words as thread_to_quad<true> + load_quad<true, true>.
__device__ __forceinline__ void load_quad_uni(const BufPlanes &mem, const LatGeom &g, const uint32_t colour, const uint32_t gid,
                                              uint32_t &Q, uint32_t &vQ, uint32_t own[4], QuadNbr &n)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t cl = uint32_t(g.cols_log2), rb = 16u << cl; // row bytes of a plane
    const uint32_t wave = __builtin_amdgcn_readfirstlane(gid >> 6), lane = gid & 63u;
    uint32_t y, col;
    bool lo_row, hi_row; // wave-uniform: the wave contains row 0 / row H-1
    uint32_t ypar;       // wave-uniform: y & 1
    if (cl >= 6) {       // a wave never leaves its row: y itself is wave-uniform
        const uint32_t ys = wave >> (cl - 6);
        y = ys;
        col = gid & ((1u << cl) - 1);
        lo_row = ys == 0;
        hi_row = ys + 1 == g.H;
        ypar = ys & 1u;
    } else {
        const uint32_t rpw = 64u >> cl, ybase = ((wave >> 1) << (7 - cl)) + (wave & 1u);
        y = ybase + 2 * (lane >> cl);
        col = lane & ((1u << cl) - 1);
        lo_row = ybase == 0;
        hi_row = ybase + 2 * (rpw - 1) + 1 >= g.H;
        ypar = wave & 1u;
    }
    Q = (y << cl) + col;
    vQ = Q << 4;
    uint32_t vU, vD;
    if (lo_row | hi_row) { // two waves per plane: per-lane wrap
        const uint32_t yu = (y == 0 ? g.H : y) - 1, yd = (y + 1 == g.H) ? 0 : y + 1;
        vU = ((yu << cl) + col) << 4;
        vD = ((yd << cl) + col) << 4;
    } else {
        vU = vQ - rb;
        vD = vQ + rb;
    }
    const bool odd = ((ypar + colour) & 1u) != 0;
    const uint32_t t16 = vQ & (rb - 1), rowb = vQ - t16;
    const uint32_t vS = rowb + (odd ? ((t16 + 16) & (rb - 1)) : ((t16 - 4) & (rb - 1)));
    const u32x4 o4 = __builtin_amdgcn_raw_buffer_load_b128(mem.rsrc, vQ, mem.own_off, 0);
    const u32x4 c4 = __builtin_amdgcn_raw_buffer_load_b128(mem.rsrc, vQ, mem.oth_off, 0);
    const u32x4 u4 = __builtin_amdgcn_raw_buffer_load_b128(mem.rsrc, vU, mem.oth_off, 0);
    const u32x4 d4 = __builtin_amdgcn_raw_buffer_load_b128(mem.rsrc, vD, mem.oth_off, 0);
    const uint32_t sw = __builtin_amdgcn_raw_buffer_load_b32(mem.rsrc, vS, mem.oth_off, 0);
    own[0] = o4.x; own[1] = o4.y; own[2] = o4.z; own[3] = o4.w;
    n.ce[0] = c4.x; n.ce[1] = c4.y; n.ce[2] = c4.z; n.ce[3] = c4.w;
    n.up[0] = u4.x; n.up[1] = u4.y; n.up[2] = u4.z; n.up[3] = u4.w;
    n.dn[0] = d4.x; n.dn[1] = d4.y; n.dn[2] = d4.z; n.dn[3] = d4.w;
    n.si[0] = sw;
    side_words(n, odd);
}

// Satisfied bonds and up spins of a colour-1 quad AFTER its update (fused energy measurement): every bond joins
// a colour-0 and a colour-1 site, so the four bonds of the colour-1 sites cover each bond once, and every
// colour-0 word is the centre word of exactly one colour-1 quad.  Same totals as lat_measure_kernel.
template <bool PMJ>
__device__ __forceinline__ void quad_measure(const uint32_t own_new[4], const QuadNbr &n, const QuadSigns &js,
                                             const uint32_t jneg_uniform, uint32_t &sat, uint32_t &up)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t a0, a1, a2, a3;
        if constexpr (PMJ) {
            a0 = own_new[q] ^ n.up[q] ^ js.w[q][0];
            a1 = own_new[q] ^ n.dn[q] ^ js.w[q][1];
            a2 = own_new[q] ^ n.ce[q] ^ js.w[q][2];
            a3 = own_new[q] ^ n.si[q] ^ js.w[q][3];
        } else {
            const uint32_t o = own_new[q] ^ jneg_uniform;
            a0 = o ^ n.up[q]; a1 = o ^ n.dn[q]; a2 = o ^ n.ce[q]; a3 = o ^ n.si[q];
        }
        sat += __popc(a0) + __popc(a1) + __popc(a2) + __popc(a3);
        // (no up-spin count: this measurement serves lattices without a field only -- E = |J| (bonds - 2 sat) -- and the
        // magnetisation is not part of the per-timestep output; `up` stays for the signature's sake)
        (void)up;
    }
}

// A quad's new words, held back by the measuring kernel until the end of the wave (see lat_sweep_measure_kernel)
struct PendingQuad {
    uint32_t w[4];
    uint32_t widx[4]; // word indices in the own plane (widx[0] * 4 = the byte offset of a vector store)
};

template <bool VEC, typename Mem>
__device__ __forceinline__ void store_pending(const Mem &mem, const PendingQuad &p)
{
    if constexpr (VEC) {
        mem.store4(p.widx[0], make_uint4(p.w[0], p.w[1], p.w[2], p.w[3]));
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) mem.store1(p.widx[q], p.w[q]);
    }
}

template <bool VEC, bool PMJ, bool UNI, typename Mem, bool MEASURE = false>
__device__ __forceinline__ void update_quad(const Mem &mem, const LatGeom &g, const uint32_t colour, const uint64_t t, const uint2 key,
                                            const PhiloxVKeys &vk, const LatThr thr, const uint32_t *__restrict__ jn,
                                            const uint32_t jneg_uniform, const uint32_t gid, uint32_t *sat = nullptr,
                                            uint32_t *up = nullptr, PendingQuad *pending = nullptr)
{
    if constexpr (VEC && UNI && std::is_same<Mem, BufPlanes>::value) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        uint32_t Q, vQ, own[4], acc[4];
        QuadNbr n;
        QuadSigns js;
        load_quad_uni(mem, g, colour, gid, Q, vQ, own, n);
        load_signs<PMJ>(jn, g, Q, js);
        const uint32_t widx[4] = {4 * Q, 4 * Q + 1, 4 * Q + 2, 4 * Q + 3};
        quad_flips<PMJ>(own, n, widx, g, colour, t, key, vk, thr, js, jneg_uniform, Q, acc);
#if defined(ISINGMC_DIAG_OLD_FUSED_STORE)
        // diagnostics only (tools/store_hazard_variants.sh): the placement rounds 1-2 shipped -- store, then the counting -- with
        // ISINGMC_DIAG_OLD_FUSED_STORE = 0: as it was; n > 0: s_nop (n - 1) behind the store; -1: immediate soffset
        if constexpr (MEASURE) {
            if constexpr (ISINGMC_DIAG_OLD_FUSED_STORE < 0)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{own[0] ^ acc[0], own[1] ^ acc[1], own[2] ^ acc[2], own[3] ^ acc[3]}, mem.rsrc,
                                                       vQ + mem.own_off, 0, 0);
            else
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{own[0] ^ acc[0], own[1] ^ acc[1], own[2] ^ acc[2], own[3] ^ acc[3]}, mem.rsrc,
                                                       vQ, mem.own_off, 0);
            uint32_t own_new[4] = {own[0] ^ acc[0], own[1] ^ acc[1], own[2] ^ acc[2], own[3] ^ acc[3]};
            // the asm reads and "writes" the four words: the counting below cannot be scheduled in front of it
            if constexpr (ISINGMC_DIAG_OLD_FUSED_STORE == 1) asm volatile("s_nop 0" : "+v"(own_new[0]), "+v"(own_new[1]), "+v"(own_new[2]), "+v"(own_new[3]));
            if constexpr (ISINGMC_DIAG_OLD_FUSED_STORE == 2) asm volatile("s_nop 1" : "+v"(own_new[0]), "+v"(own_new[1]), "+v"(own_new[2]), "+v"(own_new[3]));
            if constexpr (ISINGMC_DIAG_OLD_FUSED_STORE == 4) asm volatile("s_nop 3" : "+v"(own_new[0]), "+v"(own_new[1]), "+v"(own_new[2]), "+v"(own_new[3]));
            if constexpr (ISINGMC_DIAG_OLD_FUSED_STORE == 8) asm volatile("s_nop 7" : "+v"(own_new[0]), "+v"(own_new[1]), "+v"(own_new[2]), "+v"(own_new[3]));
            // 20 + K: the third data register is complemented K wait states behind the store and restored at once (the value the
            // counting sees is unchanged): how long is the window?
#define ISINGMC_DIAG_WINDOW(K, NOP) \
            if constexpr (ISINGMC_DIAG_OLD_FUSED_STORE == 20 + K) \
                asm volatile(NOP "v_not_b32 %2, %2\n\tv_not_b32 %2, %2" : "+v"(own_new[0]), "+v"(own_new[1]), "+v"(own_new[2]), "+v"(own_new[3]));
            ISINGMC_DIAG_WINDOW(0, "")
            ISINGMC_DIAG_WINDOW(1, "s_nop 0\n\t")
            ISINGMC_DIAG_WINDOW(2, "s_nop 1\n\t")
            ISINGMC_DIAG_WINDOW(3, "s_nop 2\n\t")
            ISINGMC_DIAG_WINDOW(4, "s_nop 3\n\t")
            ISINGMC_DIAG_WINDOW(6, "s_nop 5\n\t")
#undef ISINGMC_DIAG_WINDOW
            quad_measure<PMJ>(own_new, n, js, jneg_uniform, *sat, *up);
            pending->widx[0] = 0xFFFFFFFFu; // nothing left to store
            return;
        }
#endif
        if constexpr (MEASURE) {
            // NOT stored here: the vector store (buffer_store_dwordx4 with a register offset) reads its four data registers
            // some cycles after it issues, the compiler's hazard model inserts no wait state for that form, and the counting
            // below would overwrite them at once -- on a loaded chip a word of the quad then arrived in memory as a bit
            // count (wrong spins from ~1500 workgroups per launch on; round 3).  The kernel stores as its last instruction.
#pragma unroll
            for (int q = 0; q < 4; q++) { pending->w[q] = own[q] ^ acc[q]; pending->widx[q] = widx[q]; }
            quad_measure<PMJ>(pending->w, n, js, jneg_uniform, *sat, *up);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{own[0] ^ acc[0], own[1] ^ acc[1], own[2] ^ acc[2], own[3] ^ acc[3]}, mem.rsrc,
                                                   vQ, mem.own_off, 0);
        }
        return;
    }
    uint32_t Q, qy, qxw;
    thread_to_quad<UNI>(g, gid, Q, qy, qxw);

    uint32_t own[4], widx[4], acc[4];
    QuadNbr n;
    QuadSigns js;
    load_signs<PMJ>(jn, g, Q, js);
    load_quad<VEC, UNI>(mem, g, colour, Q, qy, qxw, own, n, widx);
    quad_flips<PMJ>(own, n, widx, g, colour, t, key, vk, thr, js, jneg_uniform, Q, acc);

    if constexpr (MEASURE) { // stored by the kernel at its end, as above
#pragma unroll
        for (int q = 0; q < 4; q++) { pending->w[q] = own[q] ^ acc[q]; pending->widx[q] = widx[q]; }
        quad_measure<PMJ>(pending->w, n, js, jneg_uniform, *sat, *up);
    } else if constexpr (VEC) {
        mem.store4(widx[0], make_uint4(own[0] ^ acc[0], own[1] ^ acc[1], own[2] ^ acc[2], own[3] ^ acc[3]));
    } else {
#pragma unroll
        for (int q = 0; q < 4; q++) mem.store1(widx[q], own[q] ^ acc[q]);
    }
}

constexpr uint32_t MEASURE_SLOTS = 16;
#ifndef ISINGMC_MEASURE_WAVES
#define ISINGMC_MEASURE_WAVES 7 // waves per SIMD the measuring kernel is compiled for (<= 72 VGPRs; it wanted 74-80 and ran at 6)
#endif

// Colour-1 half-sweep that also measures (energies after every timestep, lattice.rs:445-455): out[r * stride] +=
// satisfied bonds, out[r * stride + 1] += up spins of replica r after this timestep -- what lat_measure_kernel
// would count in a second pass over the planes.
template <bool VEC, bool PMJ, bool UNI>
__global__ __launch_bounds__(256, ISINGMC_MEASURE_WAVES) void lat_sweep_measure_kernel(
    uint32_t *__restrict__ state, const LatGeom g, const uint64_t t, const uint2 *__restrict__ keys, const LatThr thr_uniform,
    const LatThr *__restrict__ thr_replica, const uint32_t *__restrict__ jneg, const uint32_t jneg_uniform,
    unsigned long long *__restrict__ out, const size_t out_stride)
{
    // out: [replica][MEASURE_SLOTS][2] -- a workgroup adds into slot blockIdx.x % MEASURE_SLOTS, so that the
    // workgroups of a replica do not all serialise on one pair of addresses (256 per replica at 4096^2); the host sums
    __shared__ uint32_t red[2][4];
    const uint32_t colour = 1;
    const uint32_t r = blockIdx.y;
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    uint32_t sat = 0, up = 0;
    BufPlanes mem;
    mem.rsrc = __builtin_amdgcn_make_buffer_rsrc(state + size_t(r) * 2 * g.wpp, 0, int(2 * g.wpp * sizeof(uint32_t)), 0x00020000);
    mem.own_off = colour * g.wpp * 4u;
    mem.oth_off = 0;
    PendingQuad pending;
    if (gid < g.nquads) {
        const uint2 key = keys[r];
        update_quad<VEC, PMJ, UNI, BufPlanes, true>(mem, g, colour, t, key, philox_vkeys(key), thr_replica ? thr_replica[r] : thr_uniform,
                                                    PMJ ? jneg + size_t(colour) * 4 * g.wpp : nullptr, jneg_uniform, gid, &sat, &up, &pending);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sat += __shfl_xor(sat, off);
    if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = sat;
    __syncthreads();
    if (threadIdx.x == 0) { // the up-spin slot (slot + 1) stays zero: see quad_measure
        unsigned long long *slot = out + size_t(r) * out_stride + 2 * (blockIdx.x % MEASURE_SLOTS);
        atomicAdd(slot, (unsigned long long)(red[0][0] + red[0][1] + red[0][2] + red[0][3]));
    }
    // the quad's new words go out last (behind the barrier above: the compiler cannot hoist a store over it), so that nothing
    // writes the store's data registers after it has issued
#if defined(ISINGMC_DIAG_OLD_FUSED_STORE)
    if (gid < g.nquads && pending.widx[0] != 0xFFFFFFFFu) store_pending<VEC>(mem, pending);
#else
    if (gid < g.nquads) store_pending<VEC>(mem, pending);
#endif
}

template <bool VEC, bool PMJ, bool UNI>
__global__ __launch_bounds__(256) void lat_sweep_kernel(
    uint32_t *__restrict__ state, const LatGeom g, const uint32_t colour, const uint64_t t,
    const uint2 *__restrict__ keys, const LatThr thr_uniform, const LatThr *__restrict__ thr_replica,
    const uint32_t *__restrict__ jneg, const uint32_t jneg_uniform)
{
    const uint32_t r = blockIdx.y;
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= g.nquads) return;
    BufPlanes mem;
    mem.rsrc = __builtin_amdgcn_make_buffer_rsrc(state + size_t(r) * 2 * g.wpp, 0, int(2 * g.wpp * sizeof(uint32_t)), 0x00020000);
    mem.own_off = colour * g.wpp * 4u;
    mem.oth_off = (1 - colour) * g.wpp * 4u;
    const uint2 key = keys[r];
    update_quad<VEC, PMJ, UNI>(mem, g, colour, t, key, philox_vkeys(key), thr_replica ? thr_replica[r] : thr_uniform,
                               PMJ ? jneg + size_t(colour) * 4 * g.wpp : nullptr, jneg_uniform, gid);
}

// Looping variant for large launches (VEC + 2^k mapping): a thread decides `iters` quads, `stride` threads
// apart, one after the other (a rolled loop: same code size, same registers).  A wave of the one-quad kernel
// lives ~7 us and its slot then stays empty until the dispatcher has launched the next workgroup -- on average
// 6.8 of the 8 wave slots of a SIMD were occupied (SQ_WAVE_CYCLES), and the kernel loses 9 % going from 8 to 6.
// Here the slot is refilled by the wave's own next iteration, and the prologue (kernel arguments, key, round
// keys, threshold bits) is paid once per `iters` quads.  Same quads, same counters: bit-identical.
template <bool PMJ>
__global__ __launch_bounds__(256) void lat_sweep_loop_kernel(
    uint32_t *__restrict__ state, const LatGeom g, const uint32_t colour, const uint64_t t,
    const uint2 *__restrict__ keys, const LatThr thr_uniform, const LatThr *__restrict__ thr_replica,
    const uint32_t *__restrict__ jneg, const uint32_t jneg_uniform, const uint32_t iters)
{
    const uint32_t r = blockIdx.y;
    const uint32_t stride = gridDim.x * 256; // g.nquads == iters * stride (checked by the host)
    uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    BufPlanes mem;
    mem.rsrc = __builtin_amdgcn_make_buffer_rsrc(state + size_t(r) * 2 * g.wpp, 0, int(2 * g.wpp * sizeof(uint32_t)), 0x00020000);
    mem.own_off = colour * g.wpp * 4u;
    mem.oth_off = (1 - colour) * g.wpp * 4u;
    const uint2 key = keys[r];
    const PhiloxVKeys vk = philox_vkeys(key);
    const LatThr thr = thr_replica ? thr_replica[r] : thr_uniform;
    const uint32_t *jn = PMJ ? jneg + size_t(colour) * 4 * g.wpp : nullptr;
#pragma unroll 1
    for (uint32_t it = 0; it < iters; it++, gid += stride)
        update_quad<true, PMJ, true>(mem, g, colour, t, key, vk, thr, jn, jneg_uniform, gid);
}

// Random initial configuration: word w of plane c = Philox(key, (0, w>>2, c<<8, "LATI"))[w&3].
__attribute__((unused)) static __global__ __launch_bounds__(256) void lat_init_kernel(uint32_t *__restrict__ state, const LatGeom g,
                                                       const uint2 *__restrict__ keys,
                                                       const uint32_t first_replica)
{
    const uint32_t r = first_replica + blockIdx.y;
    const uint32_t Q = blockIdx.x * 256 + threadIdx.x;
    if (Q >= 2 * g.nquads) return;
    const uint32_t c = Q >= g.nquads, q = Q - c * g.nquads;
    const uint4 rnd = philox4x32_10(make_uint4(0, q, ctr2(0, c, 0), DOM_LAT_INIT), keys[r]);
    *reinterpret_cast<uint4 *>(state + size_t(r) * 2 * g.wpp + size_t(c) * g.wpp + 4 * size_t(q)) = rnd;
}

// Full recomputation of the satisfied-bond count and the up-spin count of every replica
// (get_energy, lattice.rs:208): every bond joins a colour-0 site to a colour-1 site, so the four
// bonds of the colour-0 sites cover each bond once.  out[2r] += satisfied, out[2r+1] += up spins.
constexpr uint32_t MEASURE_QUADS_PER_THREAD = 16;

template <bool VEC, bool PMJ>
__global__ __launch_bounds__(256) void lat_measure_kernel(
    const uint32_t *__restrict__ state, const LatGeom g, const uint32_t *__restrict__ jneg,
    const uint32_t jneg_uniform, unsigned long long *__restrict__ out, const size_t out_stride)
{
    // each block walks MEASURE_QUADS_PER_THREAD x 256 quads (coalesced, stride 256), reduces in the
    // wavefront with __shfl_xor and across its 4 waves through LDS: ONE atomic pair per block
    // (one pair per wave on two addresses per replica serialised the whole kernel)
    __shared__ uint32_t red[2][4];
    const uint32_t r = blockIdx.y;
    uint32_t sat = 0, up = 0;
    const uint32_t *p0 = state + size_t(r) * 2 * g.wpp;
    for (uint32_t i = 0; i < MEASURE_QUADS_PER_THREAD; i++) {
        const uint32_t gid = (blockIdx.x * MEASURE_QUADS_PER_THREAD + i) * 256 + threadIdx.x;
        if (gid >= g.nquads) break;
        uint32_t Q, qy, qxw, own[4], widx[4];
        thread_to_quad<false>(g, gid, Q, qy, qxw);
        QuadNbr n;
        load_quad<VEC, false>(PtrPlanes{const_cast<uint32_t *>(p0), p0 + g.wpp}, g, 0, Q, qy, qxw, own, n, widx);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t a0, a1, a2, a3;
            bond_masks<PMJ>(own[q], n, q, jneg, g.wpp, widx[q], jneg_uniform, a0, a1, a2, a3);
            sat += __popc(a0) + __popc(a1) + __popc(a2) + __popc(a3);
            up += __popc(own[q]) + __popc(n.ce[q]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sat += __shfl_xor(sat, off);
        up += __shfl_xor(up, off);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sat;
        red[1][threadIdx.x >> 6] = up;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(out + size_t(r) * out_stride, (unsigned long long)(red[0][0] + red[0][1] + red[0][2] + red[0][3]));
        atomicAdd(out + size_t(r) * out_stride + 1, (unsigned long long)(red[1][0] + red[1][1] + red[1][2] + red[1][3]));
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-resident kernel for small lattices (a replica's two planes fit in LDS_RESIDENT_MAX_BYTES).
// One workgroup owns one replica for `timesteps` whole timesteps: the planes are read from HBM once,
// every half-sweep runs out of LDS with a workgroup barrier between the colours, and they are written
// back once.  This removes the two kernel launches per timestep that bound small lattices (10 us per
// step whatever the size); the Philox counters depend on (quad, timestep, colour) only, so the
// configurations are bit-identical to the per-colour launches of lat_sweep_kernel.
// steps_out (optional): satisfied bonds / up spins after every timestep, [step][replica][2].
// ------------------------------------------------------------------------------------------------
constexpr uint32_t LDS_RESIDENT_MAX_BYTES = 64 * 1024;

template <bool VEC, bool PMJ>
__global__ __launch_bounds__(1024) void lat_resident_kernel(
    uint32_t *__restrict__ state, const LatGeom g, const uint64_t t0, const uint32_t timesteps,
    const uint2 *__restrict__ keys, const LatThr *__restrict__ thr_steps, const uint32_t thr_stride,
    const LatThr *__restrict__ thr_replica, const uint32_t *__restrict__ jneg, const uint32_t jneg_uniform,
    unsigned long long *__restrict__ steps_out, const uint32_t n_replicas)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t planes[]; // plane 0 then plane 1
    __shared__ uint32_t red[2][16];
    const uint32_t r = blockIdx.x, tid = threadIdx.x, nthreads = blockDim.x;
    uint32_t *mine = state + size_t(r) * 2 * g.wpp;
    for (uint32_t i = tid; i < g.wpp / 2; i += nthreads) // 2*wpp words = wpp/2 uint4
        reinterpret_cast<uint4 *>(planes)[i] = reinterpret_cast<const uint4 *>(mine)[i];
    const uint2 key = keys[r];
    const PhiloxVKeys vk = philox_vkeys(key);
    __syncthreads();
    for (uint32_t k = 0; k < timesteps; k++) {
        const LatThr thr = thr_replica ? thr_replica[r] : thr_steps[size_t(k) * thr_stride];
        for (uint32_t colour = 0; colour < 2; colour++) {
            for (uint32_t gid = tid; gid < g.nquads; gid += nthreads)
                update_quad<VEC, PMJ, false>(PtrPlanes{planes + colour * g.wpp, planes + (1 - colour) * g.wpp}, g, colour, t0 + k,
                                             key, vk, thr, PMJ ? jneg + size_t(colour) * 4 * g.wpp : nullptr, jneg_uniform,
                                             gid);
            __syncthreads();
        }
        if (steps_out) { // get_energy after every timestep (lattice.rs:454), same sums as lat_measure_kernel
            uint32_t sat = 0, up = 0;
            for (uint32_t gid = tid; gid < g.nquads; gid += nthreads) {
                uint32_t Q, qy, qxw, own[4], widx[4];
                thread_to_quad<false>(g, gid, Q, qy, qxw);
                QuadNbr n;
                load_quad<VEC, false>(PtrPlanes{planes, planes + g.wpp}, g, 0, Q, qy, qxw, own, n, widx);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t a0, a1, a2, a3;
                    bond_masks<PMJ>(own[q], n, q, jneg, g.wpp, widx[q], jneg_uniform, a0, a1, a2, a3);
                    sat += __popc(a0) + __popc(a1) + __popc(a2) + __popc(a3);
                    up += __popc(own[q]) + __popc(n.ce[q]);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                sat += __shfl_xor(sat, off);
                up += __shfl_xor(up, off);
            }
            if ((tid & 63) == 0) { red[0][tid >> 6] = sat; red[1][tid >> 6] = up; }
            __syncthreads();
            if (tid == 0) {
                unsigned long long s = 0, u = 0;
                for (uint32_t w = 0; w < (nthreads + 63) / 64; w++) { s += red[0][w]; u += red[1][w]; }
                steps_out[(size_t(k) * n_replicas + r) * 2] = s;
                steps_out[(size_t(k) * n_replicas + r) * 2 + 1] = u;
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < g.wpp / 2; i += nthreads)
        reinterpret_cast<uint4 *>(mine)[i] = reinterpret_cast<const uint4 *>(planes)[i];
}

// lattice energies from the satisfied-bond counters: E = |J| (n_bonds - 2 sat)  (exact in f64)
__attribute__((unused)) static __global__ void lat_energy_from_counts_kernel(unsigned long long *__restrict__ meas, const uint32_t n,
                                              const double jabs, const long long n_bonds, double *__restrict__ out)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) {
        out[r] = jabs * double(n_bonds - 2 * (long long)meas[2 * size_t(r)]);
        meas[2 * size_t(r)] = 0; // the counters are left zeroed for the next measurement (no memset per round)
        meas[2 * size_t(r) + 1] = 0;
    }
}

} // namespace isingmc
