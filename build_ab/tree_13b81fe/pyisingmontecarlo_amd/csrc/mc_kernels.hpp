// Multi-class checkerboard kernels: the bit-sliced Metropolis half-sweep of lattice_kernels.hpp for recognised
// lattices whose spins fall into more than two acceptance classes -- a field (uniform, or +-h from site to site: classes
// by satisfied bonds AND spin along / against the site's field), open boundaries (boundary sites have 3 or 2 bonds), both
// at once, or anisotropic couplings (classes by satisfied horizontal AND vertical bonds).  Same layout, same Philox
// counters, same plane-by-plane comparison and tie rule as DESIGN.md S3; only the class masks differ (mc_types.hpp).
// These inputs took the thread-per-site CSR path before (1.8e11 attempts/s at 4096^2).  Streaming kernel
// (one launch per colour) and LDS-resident kernel (small lattices, all timesteps of a call in one launch); the quad
// update itself is mc_quad_body.inc.
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
#define MC_RND(call) philox4x32_10(make_uint4(c0, Q, DOM_LAT_SWEEP, ctr2(t, colour, (call))), key, vk)
#include "mc_quad_body.inc"
#undef MC_RND
}

static_assert(N_PLANES == 7, "SPREAD: eight lanes per quad = 7 bit planes + the first residual call");
template <bool SPREAD>
__device__ __forceinline__ uint4 mc_rnd(const uint32_t *s_rand_words, const uint32_t gid, const uint32_t call, const uint32_t c0,
                                        const uint32_t Q, const uint64_t t, const uint32_t colour, const uint2 key, const PhiloxVKeys &vk)
{
    if constexpr (SPREAD) return reinterpret_cast<const uint4 *>(s_rand_words)[8 * gid + call];
    else return philox4x32_10(make_uint4(c0, Q, DOM_LAT_SWEEP, ctr2(t, colour, call)), key, vk);
}

// LDS-resident variant for small lattices (both planes <= LDS_RESIDENT_MAX_BYTES): one workgroup owns one replica for
// `timesteps` whole timesteps, with a workgroup barrier between the colours -- instead of two launches per timestep
// (lat_resident_kernel's scheme for the multi-class modes).  Same quads, same counters: the same configurations.
// SPREAD (lattices of <= 128 quads per colour): eight lanes per quad draw its 7 + 1 Philox calls side by side into LDS, the
// first nquads threads then decide their quads from those words -- as lat_resident_spread_kernel (spread_kernels.hpp)
template <int MODE, bool PMJ, bool FS, bool SPREAD>
__global__ __launch_bounds__(1024) void lat_mc_resident_kernel(
    uint32_t *__restrict__ state, const LatGeom g, const uint64_t t0, const uint32_t timesteps, const uint2 *__restrict__ keys,
    const LatThrMC *__restrict__ thr_steps, const uint32_t thr_stride, const LatThrMC *__restrict__ thr_replica,
    const uint32_t *__restrict__ jneg, const uint32_t jneg_uniform, const McOpen open, const uint32_t *__restrict__ fneg,
    unsigned long long *__restrict__ steps_out, const uint32_t n_replicas)
{
    // steps_out (optional): the counters of lat_mc_measure_open_kernel / lat_mc_measure_aniso_kernel after every timestep,
    // [step][replica][2] (get_energy after each step, lattice.rs:454)
    constexpr int NC = McInfo<MODE>::NC;
    extern __shared__ __attribute__((aligned(16))) uint32_t mc_planes[]; // plane 0 then plane 1
    __shared__ uint32_t lo_tab[16];
    __shared__ uint32_t red[3][16];
    const uint32_t r = blockIdx.x, tid = threadIdx.x, nthreads = blockDim.x;
    uint32_t *mine = state + size_t(r) * 2 * g.wpp;
    for (uint32_t i = tid; i < g.wpp / 2; i += nthreads) // 2 * wpp words = wpp / 2 uint4
        reinterpret_cast<uint4 *>(mc_planes)[i] = reinterpret_cast<const uint4 *>(mine)[i];
    const uint2 key = keys[r];
    const PhiloxVKeys vk = philox_vkeys(key);
    constexpr bool UNI = false; // thread_to_quad / load_quad through the LDS pointers
    const BufPlanes bmem{};     // not used without UNI
    for (uint32_t k = 0; k < timesteps; k++) {
        const LatThrMC *tp = thr_replica ? thr_replica + r : thr_steps + size_t(k) * thr_stride;
        __syncthreads(); // the planes are loaded / the previous timestep's ties have read lo_tab
        if (tid < 16) lo_tab[tid] = tid < uint32_t(NC) ? tp->lo[tid] : 0u;
        __syncthreads();
        const uint64_t t = t0 + k;
        for (uint32_t colour = 0; colour < 2; colour++) {
            const PtrPlanes mem{mc_planes + colour * g.wpp, mc_planes + (1 - colour) * g.wpp};
            if constexpr (SPREAD) { // blockDim.x >= 8 * nquads (host); Q == gid under the row-major mapping
                uint4 *s_rand = reinterpret_cast<uint4 *>(mc_planes + 2 * g.wpp);
                if ((tid >> 3) < g.nquads)
                    s_rand[tid] = philox4x32_10(make_uint4(uint32_t(t), tid >> 3, DOM_LAT_SWEEP, ctr2(t, colour, tid & 7u)), key, vk);
                __syncthreads();
            }
            for (uint32_t gid = tid; gid < g.nquads; gid += nthreads) {
#define MC_RND(call) mc_rnd<SPREAD>(mc_planes + 2 * g.wpp, gid, (call), c0, Q, t, colour, key, vk)
#include "mc_quad_body.inc"
#undef MC_RND
            }
            __syncthreads();
        }
        if (steps_out) {
            // c0 = satisfied bonds (anisotropic: the horizontal ones), c1 = up spins, c2 = the vertical satisfied bonds /
            // the spins along their site's field (sign planes): packed into the high half of word 0 as the measure kernels do
            uint32_t c0 = 0, c1 = 0, c2 = 0;
            for (uint32_t gid = tid; gid < g.nquads; gid += nthreads) {
                uint32_t Q, qy, qxw, own[4], widx[4];
                thread_to_quad<false>(g, gid, Q, qy, qxw);
                QuadNbr n;
                load_quad<true, false>(PtrPlanes{mc_planes, mc_planes + g.wpp}, g, 0, Q, qy, qxw, own, n, widx);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t a0, a1, a2, a3;
                    bond_masks<PMJ>(own[q], n, q, jneg, g.wpp, widx[q], jneg_uniform, a0, a1, a2, a3);
                    if constexpr (MODE == MC_ANISO) {
                        c2 += __popc(a0) + __popc(a1);
                        c0 += __popc(a2) + __popc(a3);
                    } else {
                        uint32_t p_up = 0xFFFFFFFFu, p_dn = 0xFFFFFFFFu, p_si = 0xFFFFFFFFu;
                        if constexpr (McInfo<MODE>::OPEN) mc_presence(g, open, 0, qy, qxw + q, p_up, p_dn, p_si);
                        c0 += __popc(a0 & p_up) + __popc(a1 & p_dn) + __popc(a2) + __popc(a3 & p_si);
                        if constexpr (FS) c2 += __popc(own[q] ^ fneg[widx[q]]) + __popc(n.ce[q] ^ fneg[g.wpp + widx[q]]);
                    }
                    c1 += __popc(own[q]) + __popc(n.ce[q]);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                c0 += __shfl_xor(c0, off);
                c1 += __shfl_xor(c1, off);
                c2 += __shfl_xor(c2, off);
            }
            if ((tid & 63) == 0) { red[0][tid >> 6] = c0; red[1][tid >> 6] = c1; red[2][tid >> 6] = c2; }
            __syncthreads();
            if (tid == 0) {
                unsigned long long s0 = 0, s1 = 0, s2 = 0;
                for (uint32_t w = 0; w < (nthreads + 63) / 64; w++) { s0 += red[0][w]; s1 += red[1][w]; s2 += red[2][w]; }
                steps_out[(size_t(k) * n_replicas + r) * 2] = s0 | (s2 << 32);
                steps_out[(size_t(k) * n_replicas + r) * 2 + 1] = s1;
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < g.wpp / 2; i += nthreads)
        reinterpret_cast<uint4 *>(mine)[i] = reinterpret_cast<const uint4 *>(mc_planes)[i];
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
