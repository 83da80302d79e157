// Lane-parallel variant of the LDS-resident lattice kernel (small lattices).  It lives in a translation unit of its own
// (spread_kernels.hip), like the strip kernel: instantiated next to the streaming kernels in isingmc.hip it changed THEIR
// register allocation (lat_sweep_loop_kernel<false>: 64 -> 66 VGPRs, one wave per SIMD less; tests/test_build_properties.py).
#pragma once
#include "lattice_kernels.hpp"
#include "spread_types.hpp"

namespace isingmc {

// Small lattices (at most 128 quads per colour, e.g. 64 x 64 ... 256 x 128): lat_resident_kernel (lattice_kernels.hpp) gives a quad to ONE lane,
// which then draws the quad's 7 + 1 Philox calls one after the other -- a chain of ~2 300 cycles per half-sweep during which
// most lanes of the (single) wave are idle.  Here EIGHT lanes serve a quad: lane 8 q + c draws call c (bit planes 0 .. 6, and
// the first residual call) and leaves its four words in LDS; after a barrier thread q (the first nquads threads: whole waves) takes quad q's 32 words and decides
// exactly as quad_flips_pre does from pre-drawn words.  Same counters, same decisions: bit-identical to that kernel.
static_assert(N_PLANES == 7, "lat_resident_spread_kernel: eight lanes per quad = 7 bit planes + the first residual call");
// LPQ = lanes per quad: 8 (one call each), 4 or 2 (two / four calls each) -- whatever lets 1024 threads cover the colour
template <bool VEC, bool PMJ, int LPQ>
__global__ __launch_bounds__(1024) void lat_resident_spread_kernel(
    uint32_t *__restrict__ state, const LatGeom g, const uint64_t t0, const uint32_t timesteps,
    const uint2 *__restrict__ keys, const LatThr *__restrict__ thr_steps, const uint32_t thr_stride,
    const LatThr *__restrict__ thr_replica, const uint32_t *__restrict__ jneg, const uint32_t jneg_uniform,
    unsigned long long *__restrict__ steps_out, const uint32_t n_replicas)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t planes[]; // plane 0, plane 1, then 32 random words per quad
    __shared__ uint32_t red[2][16];
    const uint32_t r = blockIdx.x, tid = threadIdx.x, nthreads = blockDim.x;
    uint4 *s_rand = reinterpret_cast<uint4 *>(planes + 2 * g.wpp);
    uint32_t *mine = state + size_t(r) * 2 * g.wpp;
    for (uint32_t i = tid; i < g.wpp / 2; i += nthreads)
        reinterpret_cast<uint4 *>(planes)[i] = reinterpret_cast<const uint4 *>(mine)[i];
    const uint2 key = keys[r];
    const PhiloxVKeys vk = philox_vkeys(key);
    const uint32_t quad = tid / LPQ, call = tid % LPQ; // blockDim.x >= LPQ * nquads (host)
    __syncthreads();
    for (uint32_t k = 0; k < timesteps; k++) {
        const LatThr thr = thr_replica ? thr_replica[r] : thr_steps[size_t(k) * thr_stride];
        const uint64_t t = t0 + k;
        for (uint32_t colour = 0; colour < 2; colour++) {
            if (quad < g.nquads) { // Q == the quad's index under the row-major mapping (thread_to_quad<false>)
#pragma unroll
                for (uint32_t c = 0; c < 8 / LPQ; c++)
                    s_rand[8 * quad + call + LPQ * c] = philox4x32_10(make_uint4(uint32_t(t), quad, DOM_LAT_SWEEP, ctr2(t, colour, call + LPQ * c)), key, vk);
            }
            __syncthreads();
            if (tid < g.nquads) { // the deciding lanes are the FIRST nquads threads: whole waves, not every eighth lane of all of them
                QuadRandom R;
#pragma unroll
                for (int p = 0; p < N_PLANES; p++) {
                    const uint4 v = s_rand[8 * tid + p];
                    R.rr[p][0] = v.x; R.rr[p][1] = v.y; R.rr[p][2] = v.z; R.rr[p][3] = v.w;
                }
                R.tie = s_rand[8 * tid + N_PLANES];
                const PtrPlanes mem{planes + colour * g.wpp, planes + (1 - colour) * g.wpp};
                uint32_t Q, qy, qxw, own[4], widx[4], acc[4];
                thread_to_quad<false>(g, tid, Q, qy, qxw);
                QuadNbr n;
                QuadSigns js;
                load_signs<PMJ>(PMJ ? jneg + size_t(colour) * 4 * g.wpp : nullptr, g, Q, js);
                load_quad<VEC, false>(mem, g, colour, Q, qy, qxw, own, n, widx);
                quad_flips_pre<PMJ>(own, n, widx, g, colour, t, key, vk, thr, js, jneg_uniform, Q, R, acc);
                if constexpr (VEC) {
                    mem.store4(widx[0], make_uint4(own[0] ^ acc[0], own[1] ^ acc[1], own[2] ^ acc[2], own[3] ^ acc[3]));
                } else {
#pragma unroll
                    for (int q = 0; q < 4; q++) mem.store1(widx[q], own[q] ^ acc[q]);
                }
            }
            __syncthreads();
        }
        if (steps_out) { // as lat_resident_kernel
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
                unsigned long long sum = 0, u = 0;
                for (uint32_t w = 0; w < (nthreads + 63) / 64; w++) { sum += red[0][w]; u += red[1][w]; }
                steps_out[(size_t(k) * n_replicas + r) * 2] = sum;
                steps_out[(size_t(k) * n_replicas + r) * 2 + 1] = u;
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < g.wpp / 2; i += nthreads)
        reinterpret_cast<uint4 *>(mine)[i] = reinterpret_cast<const uint4 *>(planes)[i];
}

} // namespace isingmc
