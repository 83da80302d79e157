// Multi-class checkerboard kernels: the bit-sliced Metropolis half-sweep of lattice_kernels.hpp for recognised
// lattices whose spins fall into more than two acceptance classes -- a uniform field (classes by satisfied bonds AND
// spin value) or open boundaries (boundary sites have 3 or 2 bonds).  Same layout, same Philox counters, same
// plane-by-plane comparison and tie rule as DESIGN.md S3; only the class masks differ (mc_types.hpp).  These inputs
// took the thread-per-site CSR path before (1.8e11 attempts/s at 4096^2).
#pragma once
#include "lattice_kernels.hpp"
#include "mc_types.hpp"

namespace isingmc {

// bit-sliced count of four one-bit inputs: eq2 / eq3 / eq4 = "exactly 2 / 3 / 4 of them set"
__device__ __forceinline__ void count4(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t &eq2, uint32_t &eq3, uint32_t &eq4)
{
    const uint32_t s01 = a0 ^ a1, c01 = a0 & a1, s23 = a2 ^ a3, c23 = a2 & a3;
    const uint32_t k0 = s01 ^ s23, k1 = c01 ^ c23 ^ (s01 & s23);
    eq4 = c01 & c23;
    eq3 = k1 & k0;
    eq2 = k1 & ~k0;
}

// class masks of the 32 spins of word q; open boundaries mask the bonds that do not exist.
// hole_up / hole_dn: the whole row has no upper / lower neighbour; hole_si: the one bit whose side neighbour is absent
template <int MODE>
__device__ __forceinline__ void mc_classes(const uint32_t own, const uint32_t a0, const uint32_t a1, const uint32_t a2, const uint32_t a3,
                                           const uint32_t p_up, const uint32_t p_dn, const uint32_t p_si, uint32_t mask[MC_MAX_CLASSES])
{
    if constexpr (MODE == MC_FIELD) {
        uint32_t e2, e3, e4;
        count4(a0, a1, a2, a3, e2, e3, e4);
        mask[0] = e2 & ~own; mask[1] = e2 & own;
        mask[2] = e3 & ~own; mask[3] = e3 & own;
        mask[4] = e4 & ~own; mask[5] = e4 & own;
    } else if constexpr (MODE == MC_FIELD_OPEN) {
        // `own` is sigma here (spin bit ^ field-sign bit); sat / unsat counted over the bonds that exist
        const uint32_t s0 = a0 & p_up, s1 = a1 & p_dn, s2 = a2, s3 = a3 & p_si;
        const uint32_t u0 = ~a0 & p_up, u1 = ~a1 & p_dn, u2 = ~a2, u3 = ~a3 & p_si;
        const uint32_t s01 = s0 ^ s1, c01 = s0 & s1, s23 = s2 ^ s3, c23 = s2 & s3;
        const uint32_t k0 = s01 ^ s23, k1 = c01 ^ c23 ^ (s01 & s23), e4 = c01 & c23;
        const uint32_t e1 = k0 & ~k1, e2 = k1 & ~k0, e3 = k1 & k0;
        const uint32_t us01 = u0 ^ u1, uc01 = u0 & u1, us23 = u2 ^ u3, uc23 = u2 & u3;
        const uint32_t ub0 = us01 ^ us23, ub1 = uc01 ^ uc23 ^ (us01 & us23);
        const uint32_t none = ~(u0 | u1 | u2 | u3), one = ub0 & ~ub1, two = ub1 & ~ub0;
        const uint32_t m1 = e2 & one, m2 = (e3 & one) | (e2 & none), m3 = e3 & none, m4 = e4;
        const uint32_t m0 = (e2 & two) | (e1 & one);
        mask[0] = m1 & ~own; mask[1] = m1 & own;
        mask[2] = m2 & ~own; mask[3] = m2 & own;
        mask[4] = m3 & ~own; mask[5] = m3 & own;
        mask[6] = m4 & ~own; mask[7] = m4 & own;
        mask[8] = m0 & own;
    } else if constexpr (MODE == MC_ANISO) {
        // a0 = up, a1 = down (vertical, |Jy|); a2 = centre, a3 = side (horizontal, |Jx|)
        const uint32_t kx2 = a2 & a3, kx1 = a2 ^ a3, kx0 = ~(a2 | a3), ky2 = a0 & a1, ky1 = a0 ^ a1, ky0 = ~(a0 | a1);
        mask[0] = kx2 & ky2;
        mask[1] = kx2 & ky1;
        mask[2] = kx1 & ky2;
        mask[3] = kx2 & ky0;
        mask[4] = kx0 & ky2;
    } else {
        // a0 = up, a1 = down, a2 = centre (always exists), a3 = side
        const uint32_t s0 = a0 & p_up, s1 = a1 & p_dn, s2 = a2, s3 = a3 & p_si;
        const uint32_t u0 = ~a0 & p_up, u1 = ~a1 & p_dn, u2 = ~a2, u3 = ~a3 & p_si;
        uint32_t e2, e3, e4;
        count4(s0, s1, s2, s3, e2, e3, e4);
        const uint32_t us01 = u0 ^ u1, uc01 = u0 & u1, us23 = u2 ^ u3, uc23 = u2 & u3;
        const uint32_t none = ~(u0 | u1 | u2 | u3), one = (us01 ^ us23) & ~(uc01 | uc23);
        mask[0] = e2 & one;                 // m = 1: 2 satisfied, 1 unsatisfied (a boundary site)
        mask[1] = (e3 & one) | (e2 & none); // m = 2: the bulk's k = 3, or a corner with both bonds satisfied
        mask[2] = e3 & none;                // m = 3: a boundary site with all three bonds satisfied
        mask[3] = e4;                       // m = 4
    }
}

template <int MODE>
struct McInfo {
    static constexpr int NC = MODE == MC_FIELD_OPEN ? 9 : MODE == MC_FIELD ? 6 : MODE == MC_ANISO ? 5 : 4;
    static constexpr bool FIELD = MODE == MC_FIELD || MODE == MC_FIELD_OPEN, OPEN = MODE == MC_OPEN || MODE == MC_FIELD_OPEN;
};

// presence masks of word w (global word index in the plane's row y, first word xw + q) for open boundaries
__device__ __forceinline__ void mc_presence(const LatGeom &g, const McOpen open, const uint32_t colour, const uint32_t y, const uint32_t xword,
                                            uint32_t &p_up, uint32_t &p_dn, uint32_t &p_si)
{
    p_up = (open.open_y && y == 0) ? 0u : 0xFFFFFFFFu;
    p_dn = (open.open_y && y + 1 == g.H) ? 0u : 0xFFFFFFFFu;
    p_si = 0xFFFFFFFFu;
    if (open.open_x) {
        // colour-c sites of row y sit at x = 2i + o, o = (y + c) & 1.  o = 0: the side neighbour (index i - 1) is the LEFT one,
        // absent for x = 0 (bit 0 of the row's first word); o = 1: it is the RIGHT one (index i + 1), absent for x = W - 1
        // (bit 31 of the row's last word).  The centre neighbour (same index) always exists.
        const bool odd = (y + colour) & 1u;
        if (!odd && xword == 0) p_si = ~1u;
        if (odd && xword + 1 == g.wpr) p_si = ~(1u << 31);
    }
}

// UNI: the division-free, wave-uniform thread -> quad mapping of the streaming kernels (load_quad_uni; the host
// passes it when g.cols_log2 >= 0); else thread_to_quad / load_quad.  Same quads either way.
// FS: field-sign planes (fneg), for the FIELD modes only
template <int MODE, bool PMJ, bool UNI, bool FS>
__global__ __launch_bounds__(256) void lat_mc_sweep_kernel(
    uint32_t *__restrict__ state, const LatGeom g, const uint32_t colour, const uint64_t t, const uint2 *__restrict__ keys,
    const LatThrMC thr_uniform, const LatThrMC *__restrict__ thr_replica, const uint32_t *__restrict__ jneg,
    const uint32_t jneg_uniform, const McOpen open, const uint32_t *__restrict__ fneg)
{
    constexpr int NC = McInfo<MODE>::NC;
    const uint32_t r = blockIdx.y;
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    // per-replica thresholds are wave-uniform: scalar loads
    const LatThrMC *tp = thr_replica ? thr_replica + r : &thr_uniform;
    __shared__ uint32_t lo_tab[16]; // low threshold words by class, for the tie stage (one replica per workgroup)
    if (threadIdx.x < 16) lo_tab[threadIdx.x] = threadIdx.x < uint32_t(NC) ? tp->lo[threadIdx.x] : 0u;
    __syncthreads();
    if (gid >= g.nquads) return;
    uint32_t *mine = state + size_t(r) * 2 * g.wpp;
    const PtrPlanes mem{mine + size_t(colour) * g.wpp, mine + size_t(1 - colour) * g.wpp};
    BufPlanes bmem;
    if constexpr (UNI) {
        bmem.rsrc = __builtin_amdgcn_make_buffer_rsrc(mine, 0, int(2 * g.wpp * sizeof(uint32_t)), 0x00020000);
        bmem.own_off = colour * g.wpp * 4u;
        bmem.oth_off = (1 - colour) * g.wpp * 4u;
    }
    const uint2 key = keys[r];
    const PhiloxVKeys vk = philox_vkeys(key);
    uint32_t hi[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) hi[c] = __builtin_amdgcn_readfirstlane(tp->hi[c]);
    const uint32_t costly = __builtin_amdgcn_readfirstlane(tp->costly);

    uint32_t Q, qy, qxw, own[4], widx[4], vQ = 0;
    QuadNbr n;
    QuadSigns js;
    if constexpr (UNI) {
        load_quad_uni(bmem, g, colour, gid, Q, vQ, own, n);
        qy = Q >> uint32_t(g.cols_log2);
        qxw = (Q & ((1u << uint32_t(g.cols_log2)) - 1)) * 4;
#pragma unroll
        for (int q = 0; q < 4; q++) widx[q] = 4 * Q + q;
        load_signs<PMJ>(PMJ ? jneg + size_t(colour) * 4 * g.wpp : nullptr, g, Q, js);
    } else {
        thread_to_quad<false>(g, gid, Q, qy, qxw);
        load_signs<PMJ>(PMJ ? jneg + size_t(colour) * 4 * g.wpp : nullptr, g, Q, js);
        load_quad<true, false>(mem, g, colour, Q, qy, qxw, own, n, widx);
    }

    uint32_t mask[4][NC], lt[4], und[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t a0, a1, a2, a3, m6[MC_MAX_CLASSES];
        if constexpr (PMJ) {
            a0 = own[q] ^ n.up[q] ^ js.w[q][0];
            a1 = own[q] ^ n.dn[q] ^ js.w[q][1];
            a2 = own[q] ^ n.ce[q] ^ js.w[q][2];
            a3 = own[q] ^ n.si[q] ^ js.w[q][3];
        } else {
            bond_masks<false>(own[q], n, q, nullptr, g.wpp, widx[q], jneg_uniform, a0, a1, a2, a3);
        }
        uint32_t p_up = 0xFFFFFFFFu, p_dn = 0xFFFFFFFFu, p_si = 0xFFFFFFFFu;
        if constexpr (McInfo<MODE>::OPEN) mc_presence(g, open, colour, qy, qxw + q, p_up, p_dn, p_si);
        uint32_t sigma = own[q]; // the spin bit, or spin x sign of the site's field
        if constexpr (McInfo<MODE>::FIELD) {
            if constexpr (FS) sigma ^= fneg[size_t(colour) * g.wpp + widx[q]];
            else if constexpr (MODE == MC_FIELD_OPEN) sigma ^= open.fneg_uniform;
        }
        mc_classes<MODE>(sigma, a0, a1, a2, a3, p_up, p_dn, p_si, m6);
        und[q] = 0;
        lt[q] = 0;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            mask[q][c] = ((costly >> c) & 1u) ? m6[c] : 0u; // wave-uniform: a class that flips outright needs no random number
            und[q] |= mask[q][c];
        }
    }
    uint32_t acc[4];
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = ~und[q]; // every spin outside the costly classes flips

    const uint32_t c0 = uint32_t(t);
#pragma unroll
    for (int p = N_PLANES - 1; p >= 0; p--) { // least significant plane first, as quad_planes
        const uint4 rnd = philox4x32_10(make_uint4(c0, Q, DOM_LAT_SWEEP, ctr2(t, colour, p)), key, vk);
        const uint32_t rr[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
        uint32_t tbw[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < NC; c++)
            if ((hi[c] >> (N_PLANES - 1 - p)) & 1u) { // scalar branch: the threshold bits are per replica
#pragma unroll
                for (int q = 0; q < 4; q++) tbw[q] |= mask[q][c];
            }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            lt[q] = __builtin_amdgcn_bitop3_b32(rr[q], tbw[q], lt[q], 0x8E);   // (~r & tb) | (~(r ^ tb) & lt)
            und[q] = __builtin_amdgcn_bitop3_b32(und[q], rr[q], tbw[q], 0x90); // eq & ~(r ^ tb)
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] |= lt[q];
#ifdef ISINGMC_TIMING_ONLY_NO_TIES // diagnostic build: what the tie stage costs (results are wrong without it)
    if (false) {
#else
    if (und[0] | und[1] | und[2] | und[3]) { // ties: the n-th of the quad in (word, bit) order takes word n % 4 of call N_PLANES + n / 4
#endif
        uint32_t nres = 0;
        uint4 w = philox4x32_10(make_uint4(c0, Q, DOM_LAT_SWEEP, ctr2(t, colour, N_PLANES)), key, vk);
        // the class of a tied spin as a 3-bit index from three bit-planes per word; its threshold's low word from a
        // table in LDS (one read instead of a select per class)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t m = und[q];
            if (!m) continue;
            uint32_t k0 = mask[q][1] | mask[q][3], k1 = mask[q][2] | mask[q][3], k2 = 0, k3 = 0;
            if constexpr (NC > 4) { k2 = mask[q][4]; }
            if constexpr (NC > 5) { k0 |= mask[q][5]; k2 |= mask[q][5]; }
            if constexpr (NC > 8) { k1 |= mask[q][6] | mask[q][7]; k2 |= mask[q][6] | mask[q][7]; k0 |= mask[q][7]; k3 = mask[q][8]; }
            while (m) {
                const uint32_t b = __ffs(m) - 1;
                m &= m - 1;
                if (nres != 0 && (nres & 3u) == 0)
                    w = philox4x32_10(make_uint4(c0, Q, DOM_LAT_SWEEP, ctr2(t, colour, N_PLANES + (nres >> 2))), key, vk);
                const uint32_t idx = ((k0 >> b) & 1u) | (((k1 >> b) & 1u) << 1) | (((k2 >> b) & 1u) << 2) | (((k3 >> b) & 1u) << 3);
                const uint32_t lo_c = lo_tab[idx];
                if (sel4(w, nres & 3u) < lo_c) acc[q] |= 1u << b;
                nres++;
            }
        }
    }
    const uint4 out = make_uint4(own[0] ^ acc[0], own[1] ^ acc[1], own[2] ^ acc[2], own[3] ^ acc[3]);
    if constexpr (UNI) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{out.x, out.y, out.z, out.w}, bmem.rsrc, vQ, bmem.own_off, 0);
    } else {
        mem.store4(widx[0], out);
    }
}

// lat_measure_kernel with the satisfied horizontal and vertical bonds counted apart (they carry different |J|)
template <bool PMJ>
__global__ __launch_bounds__(256) void lat_mc_measure_aniso_kernel(
    const uint32_t *__restrict__ state, const LatGeom g, const uint32_t *__restrict__ jneg, const uint32_t jneg_uniform,
    unsigned long long *__restrict__ out, const size_t out_stride)
{
    __shared__ uint32_t red[3][4];
    const uint32_t r = blockIdx.y;
    uint32_t satx = 0, saty = 0, up = 0;
    const uint32_t *p0 = state + size_t(r) * 2 * g.wpp;
    for (uint32_t i = 0; i < MEASURE_QUADS_PER_THREAD; i++) {
        const uint32_t gid = (blockIdx.x * MEASURE_QUADS_PER_THREAD + i) * 256 + threadIdx.x;
        if (gid >= g.nquads) break;
        uint32_t Q, qy, qxw, own[4], widx[4];
        thread_to_quad<false>(g, gid, Q, qy, qxw);
        QuadNbr n;
        load_quad<true, false>(PtrPlanes{const_cast<uint32_t *>(p0), p0 + g.wpp}, g, 0, Q, qy, qxw, own, n, widx);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t a0, a1, a2, a3;
            bond_masks<PMJ>(own[q], n, q, jneg, g.wpp, widx[q], jneg_uniform, a0, a1, a2, a3);
            saty += __popc(a0) + __popc(a1);
            satx += __popc(a2) + __popc(a3);
            up += __popc(own[q]) + __popc(n.ce[q]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        satx += __shfl_xor(satx, off);
        saty += __shfl_xor(saty, off);
        up += __shfl_xor(up, off);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = satx;
        red[1][threadIdx.x >> 6] = saty;
        red[2][threadIdx.x >> 6] = up;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long sx = (unsigned long long)red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const unsigned long long sy = (unsigned long long)red[1][0] + red[1][1] + red[1][2] + red[1][3];
        atomicAdd(out + size_t(r) * out_stride, sx | (sy << 32)); // a lattice of < 2^32 spins: the halves cannot carry into each other
        atomicAdd(out + size_t(r) * out_stride + 1, (unsigned long long)(red[2][0] + red[2][1] + red[2][2] + red[2][3]));
    }
}

// lat_measure_kernel with the open boundaries' missing bonds left out (colour-0 sites, their four directions)
template <bool PMJ>
__global__ __launch_bounds__(256) void lat_mc_measure_open_kernel(
    const uint32_t *__restrict__ state, const LatGeom g, const uint32_t *__restrict__ jneg, const uint32_t jneg_uniform, const McOpen open,
    const uint32_t *__restrict__ fneg, unsigned long long *__restrict__ out, const size_t out_stride)
{
    __shared__ uint32_t red[3][4];
    const uint32_t r = blockIdx.y;
    uint32_t sat = 0, up = 0, along = 0; // along: spins pointing along their site's field (field-sign planes only)
    const uint32_t *p0 = state + size_t(r) * 2 * g.wpp;
    for (uint32_t i = 0; i < MEASURE_QUADS_PER_THREAD; i++) {
        const uint32_t gid = (blockIdx.x * MEASURE_QUADS_PER_THREAD + i) * 256 + threadIdx.x;
        if (gid >= g.nquads) break;
        uint32_t Q, qy, qxw, own[4], widx[4];
        thread_to_quad<false>(g, gid, Q, qy, qxw);
        QuadNbr n;
        load_quad<true, false>(PtrPlanes{const_cast<uint32_t *>(p0), p0 + g.wpp}, g, 0, Q, qy, qxw, own, n, widx);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t a0, a1, a2, a3, p_up, p_dn, p_si;
            bond_masks<PMJ>(own[q], n, q, jneg, g.wpp, widx[q], jneg_uniform, a0, a1, a2, a3);
            mc_presence(g, open, 0, qy, qxw + q, p_up, p_dn, p_si);
            sat += __popc(a0 & p_up) + __popc(a1 & p_dn) + __popc(a2) + __popc(a3 & p_si);
            up += __popc(own[q]) + __popc(n.ce[q]);
            if (fneg) along += __popc(own[q] ^ fneg[widx[q]]) + __popc(n.ce[q] ^ fneg[g.wpp + widx[q]]); // the centre word: same index, colour 1
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sat += __shfl_xor(sat, off);
        up += __shfl_xor(up, off);
        along += __shfl_xor(along, off);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = sat;
        red[1][threadIdx.x >> 6] = up;
        red[2][threadIdx.x >> 6] = along;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long s4 = (unsigned long long)red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const unsigned long long a4 = (unsigned long long)red[2][0] + red[2][1] + red[2][2] + red[2][3];
        atomicAdd(out + size_t(r) * out_stride, s4 | (a4 << 32)); // fewer than 2^31 spins with sign planes: the halves cannot carry
        atomicAdd(out + size_t(r) * out_stride + 1, (unsigned long long)(red[1][0] + red[1][1] + red[1][2] + red[1][3]));
    }
}

} // namespace isingmc
