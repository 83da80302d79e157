// Replica-packed sweep for graphs whose real sites all have ONE degree D (3..6): the 3-d cubic lattice of BASELINE
// config c5, triangular / honeycomb / random regular graphs -- ferromagnetic, antiferromagnetic, or +-J (PMJ).
// Same layout, same random numbers, same decisions as pk_sweep_kernel (packed_kernels.hpp) -- the oracle's engine D --
// with everything that kernel derives per lane and per slot from the ELL entries (is the slot used, the bond's sign, the
// degree, its parity, the class thresholds' table rows) folded into template constants and scalars:
//   * satisfied bonds: one v_bitop3 per neighbour (s ^ n ^ sign), counted with a carry-save adder tree
//     (8 three-input instructions for 6 neighbours) instead of a serial 3-bit counter (5 per neighbour);
//   * the costly classes k = D/2 + 1 .. D are compile-time patterns of the counter: one v_bitop3 each;
//   * bit-plane comparison from the least significant plane up (two v_bitop3 per word and plane, lattice_kernels.hpp);
//     the threshold word of a plane is OR_j (class_j & T_j) with T_j a scalar -- or, when every replica of the launch
//     has the same beta (UB), one scalar branch per plane picks the classes whose threshold bit is set;
//   * neighbour addresses: own offset + a scalar for translation blocks, one shift for table entries.
//   * couplings of both signs (PMJ: the 3-d +-J spin glass): the block headers of this kernel describe translations
//     whatever the signs, and carry the 64 signs of a block's slot as one 64-bit mask -- a scalar load and one
//     v_cndmask per neighbour instead of a 256-byte table read.
// Positions [class_begin, class_end) must be real sites (no padding): the host sends a class's last, padded 256-block
// through pk_sweep_kernel.
#pragma once
#include "lattice_kernels.hpp"
#include "packed_types.hpp"

namespace isingmc {

// [count == K] of the bit-sliced counter (c0 = LSB)
template <int K>
__device__ __forceinline__ uint32_t pku_match(const uint32_t c0, const uint32_t c1, const uint32_t c2)
{
    return __builtin_amdgcn_bitop3_b32(c0, c1, c2, 1 << (4 * (K & 1) + 2 * ((K >> 1) & 1) + ((K >> 2) & 1)));
}

__device__ __forceinline__ uint32_t pku_xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t pku_maj3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }

// number of set bits among D words, per bit lane: carry-save adders
template <int D>
__device__ __forceinline__ void pku_count(const uint32_t (&s)[PK_MAX_DEG], uint32_t &c0, uint32_t &c1, uint32_t &c2)
{
    static_assert(D >= 3 && D <= 6, "degree");
    const uint32_t s1 = pku_xor3(s[0], s[1], s[2]), k1 = pku_maj3(s[0], s[1], s[2]);
    if constexpr (D == 3) {
        c0 = s1;
        c1 = k1;
        c2 = 0;
    } else if constexpr (D == 4) {
        c0 = s1 ^ s[3];
        const uint32_t k2 = s1 & s[3];
        c1 = k1 ^ k2;
        c2 = k1 & k2;
    } else if constexpr (D == 5) {
        c0 = pku_xor3(s1, s[3], s[4]);
        const uint32_t k2 = pku_maj3(s1, s[3], s[4]);
        c1 = k1 ^ k2;
        c2 = k1 & k2;
    } else {
        const uint32_t s2 = pku_xor3(s[3], s[4], s[5]), k2 = pku_maj3(s[3], s[4], s[5]);
        c0 = s1 ^ s2;
        const uint32_t k3 = s1 & s2;
        c1 = pku_xor3(k1, k2, k3);
        c2 = pku_maj3(k1, k2, k3);
    }
}

// the costly classes of degree D: j = 0 .. NJ-1 <-> k_j = D/2 + 1 + j satisfied bonds, m_j = 2 k_j - D, table row m_j - 1
template <int D>
struct PkuClasses {
    static constexpr int NJ = D - D / 2;
    static constexpr int k(int j) { return D / 2 + 1 + j; }
    static constexpr int row(int j) { return 2 * k(j) - D - 1; }
};

// lane l: ~0 where bit l of the wave-uniform mask is clear (J < 0: satisfied when the spins agree), 0 where it is set
__device__ __forceinline__ uint32_t pku_neg_of(const uint2 mask)
{
    const uint64_t m = (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(mask.y))) << 32) | uint32_t(__builtin_amdgcn_readfirstlane(mask.x));
    uint32_t neg;
    asm("v_cndmask_b32 %0, -1, 0, %1" : "=v"(neg) : "s"(m));
    return neg;
}

#ifndef ISINGMC_PKU_WAVES
#define ISINGMC_PKU_WAVES 8 // workgroups of 256 threads per CU the kernel is compiled for (A/B builds: 6 or 5 + ISINGMC_PKU_VKEYS)
#endif
#ifndef ISINGMC_PKU_PRE
#define ISINGMC_PKU_PRE 0
#endif

#ifdef ISINGMC_PKU_VKEYS // A/B build: round keys 4-10 in vector registers (14 VGPRs; spills at the 64-VGPR cap of 8 waves)
#define PKU_PHILOX(c) philox4x32_10(c, key, vk)
#else
#define PKU_PHILOX(c) philox4x32_10(c, key)
#endif

// memory phase of the position-quad led by p0.  All 24 block headers first (scalar loads, one wait), then straight-line code: the
// neighbour's position is own position + the header's shift; where a block is not a translation a branch holding nothing but a
// load overwrites it with the table entry (a use of the loaded value inside the branch, or a header load per slot, makes the wave
// wait for memory once per slot); then the gathers (the shift to a byte offset drops the sign bit).
// (block headers, sign masks: read through the CONSTANT address space, so that the loads stay scalar inside the looping kernel
//  too -- there the compiler cannot prove that the kernel's own stores leave them alone and would fall back to vector loads)
typedef const __attribute__((address_space(4))) uint2 pku_const_uint2;
__device__ __forceinline__ pku_const_uint2 *pku_const(const uint2 *p) { return (pku_const_uint2 *)(uintptr_t)p; }

// Neighbour positions of the position-quad led by p0 from its 24 block headers (scalar loads, one wait): own position + the
// header's shift; + the exception of a block that is a translation for every lane but one; or -- anything else -- the table entry:
// a branch holding NOTHING but the load (a use of the loaded value inside the branch, or a header load per slot, would make the wave
// wait for memory once per slot; this way the first gather waits once for all of them, and for nothing on lattice-like graphs).
// TABLE = false: the host has checked that no block of the launch needs its table entries (every (block, slot) is a translation,
// for all lanes or for all but one): no load, no wait, nothing for the compiler to be careful about.
template <int D, bool TABLE>
__device__ __forceinline__ void pku_resolve(const __amdgpu_buffer_rsrc_t ell_rsrc, const PkUniHeaders &H, const uint32_t n_pos, const uint32_t p0,
                                            const uint32_t lane, uint32_t (&ent)[4][PK_MAX_DEG])
{
    uint2 h[4][PK_MAX_DEG];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        pku_const_uint2 *hdr = pku_const(H.shift) + size_t(__builtin_amdgcn_readfirstlane((p0 + 64 * q) >> 6)) * PK_MAX_DEG; // wave-uniform
#pragma unroll
        for (int i = 0; i < D; i++) h[q][i] = make_uint2(hdr[i].x, hdr[i].y);
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < D; i++) ent[q][i] = p0 + 64 * q + uint32_t(__builtin_amdgcn_readfirstlane(h[q][i].y));
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < D; i++) {
            const uint32_t hx = __builtin_amdgcn_readfirstlane(h[q][i].x);
            if (hx != PK_HDR_UNIFORM) {
                asm volatile(""); // (a real scalar branch, skipped by nearly every slot: if-converted, each slot pays ten instructions)
                if (!TABLE || (hx & 3u) == PK_HDR_UNIFORM_BUT_ONE) ent[q][i] += lane == ((hx >> 2) & 63u) ? uint32_t(int32_t(hx) >> 8) : 0u;
                else ent[q][i] = __builtin_amdgcn_raw_buffer_load_b32(ell_rsrc, 4 * (uint32_t(i) * n_pos + p0 + 64 * q), 0, 0);
            }
        }
}

// memory phase of the position-quad led by p0: neighbour positions, the gathers (the shift to a byte offset drops the sign bit of a
// table entry), and the own words LAST -- the wait the compiler puts in front of the first gather (a table entry may be pending)
// then finds nothing of ours in flight.
template <int D, bool TABLE>
__device__ __forceinline__ void pku_load(const __amdgpu_buffer_rsrc_t st_rsrc, const __amdgpu_buffer_rsrc_t ell_rsrc, const PkUniHeaders &H,
                                         const uint32_t n_pos, const uint32_t p0, const uint32_t lane, uint32_t (&own)[4],
                                         uint32_t (&nb)[4][PK_MAX_DEG])
{
    uint32_t ent[4][PK_MAX_DEG];
    if constexpr (!TABLE) { // nothing to wait for before the gathers: the own words go first (they need no header)
#pragma unroll
        for (int q = 0; q < 4; q++) own[q] = __builtin_amdgcn_raw_buffer_load_b32(st_rsrc, 4 * (p0 + 64 * q), 0, 0);
    }
    pku_resolve<D, TABLE>(ell_rsrc, H, n_pos, p0, lane, ent);
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int q = 0; q < 4; q++) nb[q][i] = __builtin_amdgcn_raw_buffer_load_b32(st_rsrc, ent[q][i] << 2, 0, 0);
    if constexpr (TABLE) {
#pragma unroll
        for (int q = 0; q < 4; q++) own[q] = __builtin_amdgcn_raw_buffer_load_b32(st_rsrc, 4 * (p0 + 64 * q), 0, 0);
    }
}

// classes of the position-quad led by p0 from its own and neighbour words: eq[q][j] = the replicas with k_j satisfied bonds,
// sure = flips whatever the random numbers say, und = costly and not yet decided
template <int D, bool PMJ>
__device__ __forceinline__ void pku_classes(const PkUniHeaders &H, const uint32_t p0, const uint32_t *__restrict__ tab, const uint32_t (&own)[4],
                                            const uint32_t (&nb)[4][PK_MAX_DEG], uint32_t (&eq)[4][3], uint32_t (&sure)[4], uint32_t (&und)[4])
{
    using CL = PkuClasses<D>;
    constexpr int NJ = CL::NJ;
    uint32_t all_j[3];
#pragma unroll
    for (int j = 0; j < NJ; j++) all_j[j] = tab[PK_TAB_ALL + CL::row(j)];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t sat[PK_MAX_DEG], c0, c1, c2;
        pku_const_uint2 *signs = pku_const(H.sign) + size_t(__builtin_amdgcn_readfirstlane((p0 + 64 * q) >> 6)) * PK_MAX_DEG;
#pragma unroll
        for (int i = 0; i < D; i++) // J > 0: satisfied when the spins differ
            sat[i] = pku_xor3(own[q], nb[q][i], PMJ ? pku_neg_of(make_uint2(signs[i].x, signs[i].y)) : H.negmask);
        pku_count<D>(sat, c0, c1, c2);
        eq[q][0] = pku_match<CL::k(0)>(c0, c1, c2);
        eq[q][1] = pku_match<CL::k(1)>(c0, c1, c2);
        eq[q][2] = NJ > 2 ? pku_match<CL::k(2 % NJ)>(c0, c1, c2) : 0u;
        uint32_t allm = eq[q][0] & all_j[0];
        allm = __builtin_amdgcn_bitop3_b32(eq[q][1], all_j[1], allm, 0xEA); // (a & b) | c
        if constexpr (NJ > 2) allm = __builtin_amdgcn_bitop3_b32(eq[q][2], all_j[2], allm, 0xEA);
        const uint32_t costly = NJ > 2 ? __builtin_amdgcn_bitop3_b32(eq[q][0], eq[q][1], eq[q][2], 0xFE) : eq[q][0] | eq[q][1];
        sure[q] = ~costly | allm;
        und[q] = costly & ~allm;
    }
}

// the random half: bit-planes of the acceptance uniform against the classes' thresholds, then the ties; acc[q] = the replica bits that flip
template <int D, bool UB>
__device__ __forceinline__ void pku_random(const uint32_t p0, const uint64_t t, const uint2 key,
#ifdef ISINGMC_PKU_VKEYS
                                           const PhiloxVKeys &vk,
#endif
                                           const uint32_t *__restrict__ tab, const uint32_t (&eq)[4][3], const uint32_t (&sure)[4],
                                           uint32_t (&und)[4], uint32_t (&acc)[4])
{
    using CL = PkuClasses<D>;
    constexpr int NJ = CL::NJ;
    // The random words do not depend on the spins: the first ISINGMC_PKU_PRE planes are drawn here, between the issue of the
    // gathers and the first use of their results, so that the memory round trip is covered by arithmetic of the wave's own
    const uint32_t c0w = uint32_t(t), c1w = p0;
    uint4 pre[ISINGMC_PKU_PRE > 0 ? ISINGMC_PKU_PRE : 1];
#pragma unroll
    for (int k = 0; k < ISINGMC_PKU_PRE; k++) {
        pre[k] = PKU_PHILOX(make_uint4(c0w, c1w, DOM_PK_SWEEP, ctr2(t, 0, N_PLANES - 1 - k)));
        asm volatile("" : "+v"(pre[k].x), "+v"(pre[k].y), "+v"(pre[k].z), "+v"(pre[k].w));
    }
    uint32_t lt[4] = {0, 0, 0, 0};
    uint32_t selw[2] = {0, 0}; // UB: which classes' threshold bits are set in each plane, from two table words (not 21 loads)
    if constexpr (UB) { selw[0] = tab[PK_TAB_SEL]; selw[1] = tab[PK_TAB_SEL + 1]; }
    // bit-planes, least significant first: lt' = (~r & tb) | (~(r ^ tb) & lt), und' = und & ~(r ^ tb)
#pragma unroll
    for (int pl = N_PLANES - 1; pl >= 0; pl--) {
        const uint4 rnd = N_PLANES - 1 - pl < ISINGMC_PKU_PRE ? pre[N_PLANES - 1 - pl < ISINGMC_PKU_PRE ? N_PLANES - 1 - pl : 0]
                                                              : PKU_PHILOX(make_uint4(c0w, c1w, DOM_PK_SWEEP, ctr2(t, 0, pl)));
        const uint32_t rr[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
        uint32_t T[3] = {0, 0, 0};
        if constexpr (UB) { // bit row * N_PLANES + pl of the two selection words: is this plane's bit of the row's threshold set?
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int idx = CL::row(j) * N_PLANES + pl;
                T[j] = 0u - ((selw[idx >> 5] >> (idx & 31)) & 1u);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; j++) T[j] = tab[PK_TAB_TBW + CL::row(j) * N_PLANES + pl];
        }
        if constexpr (UB) {
            // every T_j is 0 or ~0: one scalar branch selects the classes whose threshold has this bit set
            const uint32_t sel = __builtin_amdgcn_readfirstlane((T[0] & 1u) | (T[1] & 2u) | (NJ > 2 ? (T[2] & 4u) : 0u));
            const auto step = [&](auto tb_of) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t tb = tb_of(q);
                    lt[q] = __builtin_amdgcn_bitop3_b32(rr[q], tb, lt[q], 0x8E);
                    und[q] = __builtin_amdgcn_bitop3_b32(und[q], rr[q], tb, 0x90);
                }
            };
            switch (sel) {
            case 0:
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    lt[q] &= ~rr[q];
                    und[q] &= ~rr[q];
                }
                break;
            case 1: step([&](int q) { return eq[q][0]; }); break;
            case 2: step([&](int q) { return eq[q][1]; }); break;
            case 3: step([&](int q) { return eq[q][0] | eq[q][1]; }); break;
            case 4: step([&](int q) { return eq[q][2]; }); break;
            case 5: step([&](int q) { return eq[q][0] | eq[q][2]; }); break;
            case 6: step([&](int q) { return eq[q][1] | eq[q][2]; }); break;
            default: step([&](int q) { return __builtin_amdgcn_bitop3_b32(eq[q][0], eq[q][1], eq[q][2], 0xFE); }); break;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t tb = eq[q][0] & T[0];
                tb = __builtin_amdgcn_bitop3_b32(eq[q][1], T[1], tb, 0xEA);
                if constexpr (NJ > 2) tb = __builtin_amdgcn_bitop3_b32(eq[q][2], T[2], tb, 0xEA);
                lt[q] = __builtin_amdgcn_bitop3_b32(rr[q], tb, lt[q], 0x8E);
                und[q] = __builtin_amdgcn_bitop3_b32(und[q], rr[q], tb, 0x90);
            }
        }
    }

#pragma unroll
    for (int q = 0; q < 4; q++) acc[q] = sure[q] | lt[q];
    if (und[0] | und[1] | und[2] | und[3]) { // ties: n-th of the position-quad takes word n%4 of call N_PLANES + n/4
        uint32_t nres = 0;
        uint4 rnd = PKU_PHILOX(make_uint4(c0w, c1w, DOM_PK_SWEEP, ctr2(t, 0, N_PLANES)));
        uint32_t lo_j[3] = {0, 0, 0};
        if constexpr (UB) {
#pragma unroll
            for (int j = 0; j < NJ; j++) lo_j[j] = tab[PK_TAB_LO + CL::row(j) * 32];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint32_t mm = und[q];
            while (mm) {
                const uint32_t b = __ffs(mm) - 1;
                mm &= mm - 1;
                if (nres != 0 && (nres & 3u) == 0)
                    rnd = PKU_PHILOX(make_uint4(c0w, c1w, DOM_PK_SWEEP, ctr2(t, 0, N_PLANES + (nres >> 2))));
                const bool is0 = (eq[q][0] >> b) & 1u, is1 = (eq[q][1] >> b) & 1u;
                uint32_t lo;
                if constexpr (UB) lo = is0 ? lo_j[0] : is1 ? lo_j[1] : lo_j[2];
                else lo = tab[PK_TAB_LO + (is0 ? CL::row(0) : is1 ? CL::row(1) : CL::row(2 % NJ)) * 32 + b];
                if (sel4(rnd, nres & 3u) < lo) acc[q] |= 1u << b;
                nres++;
            }
        }
    }
}


// decisions of the position-quad led by p0 from its own and neighbour words; acc[q] = the replica bits that flip
template <int D, bool UB, bool PMJ>
__device__ __forceinline__ void pku_decide(const PkUniHeaders &H, const uint32_t p0, const uint64_t t, const uint2 key,
#ifdef ISINGMC_PKU_VKEYS
                                           const PhiloxVKeys &vk,
#endif
                                           const uint32_t *__restrict__ tab, const uint32_t (&own)[4], const uint32_t (&nb)[4][PK_MAX_DEG],
                                           uint32_t (&acc)[4])
{
    uint32_t eq[4][3], sure[4], und[4];
    pku_classes<D, PMJ>(H, p0, tab, own, nb, eq, sure, und);
#ifdef ISINGMC_PKU_VKEYS
    pku_random<D, UB>(p0, t, key, vk, tab, eq, sure, und, acc);
#else
    pku_random<D, UB>(p0, t, key, tab, eq, sure, und, acc);
#endif
}

template <int D, bool UB, bool PMJ, bool TABLE>
__global__ __launch_bounds__(256, ISINGMC_PKU_WAVES) void pk_sweep_uni_kernel(uint32_t *__restrict__ state, const PkGraphDev G, const PkUniHeaders H,
                                                               const uint32_t class_begin, const uint32_t class_end,
                                                               const uint64_t t, const uint2 *__restrict__ group_keys,
                                                               const uint32_t *__restrict__ tabs, const uint32_t tab_stride)
{
    const uint32_t g = blockIdx.y;
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x; // wave w of the class owns positions [256 w, 256 w + 256)
    const uint32_t p0 = class_begin + 256 * (tid >> 6) + (tid & 63u); // the quad's leader
    if (p0 >= class_end) return;
    uint32_t *st = state + size_t(g) * G.n_pos;
    const uint32_t *tab = tabs + size_t(g) * tab_stride;
    const uint2 key = group_keys[g];
#ifdef ISINGMC_PKU_VKEYS
    const PhiloxVKeys vk = philox_vkeys(key);
#define PKU_VK vk,
#else
#define PKU_VK
#endif
    const __amdgpu_buffer_rsrc_t st_rsrc = __builtin_amdgcn_make_buffer_rsrc(st, 0, int(G.n_pos * sizeof(uint32_t)), 0x00020000);
    const __amdgpu_buffer_rsrc_t ell_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(G.nbr_ell), 0, int(uint32_t(PK_MAX_DEG) * G.n_pos * uint32_t(sizeof(uint32_t))), 0x00020000);

    uint32_t own[4], nb[4][PK_MAX_DEG], acc[4];
    pku_load<D, TABLE>(st_rsrc, ell_rsrc, H, G.n_pos, p0, tid & 63u, own, nb);
    pku_decide<D, UB, PMJ>(H, p0, t, key, PKU_VK tab, own, nb, acc);
#pragma unroll
    for (int q = 0; q < 4; q++) __builtin_amdgcn_raw_buffer_store_b32(own[q] ^ acc[q], st_rsrc, 4 * (p0 + 64 * q), 0, 0);
#undef PKU_VK
}

#undef PKU_PHILOX

} // namespace isingmc
