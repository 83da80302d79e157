// Philox4x32-10 counter-based RNG for gfx950 device code (Salmon et al., SC'11).
// One call = 10 rounds of two 32x32->64 multiplies (v_mad_u64_u32 / v_mul_hi_u32) + xors; the key
// schedule is wave-uniform (the key is per replica) and stays on the scalar unit.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace isingmc {

constexpr uint32_t DOM_LAT_SWEEP = 0x4C415453u; // "LATS"
constexpr uint32_t DOM_LAT_INIT = 0x4C415449u;  // "LATI"
constexpr uint32_t DOM_GEN_SWEEP = 0x47454E53u; // "GENS"
constexpr uint32_t DOM_GEN_INIT = 0x47454E49u;  // "GENI"

// One Philox round.  VECTOR_XOR3: hi ^ ctr ^ key as ONE v_bitop3_b32 (truth table 0x96); hipcc emits two
// v_xor for the plain expression.  The builtin pins its operands to the vector unit, so it is used only
// where the operands are lane-varying anyway (see below).
template <bool VECTOR_XOR3>
__device__ __forceinline__ void philox_round(uint4 &c, uint2 &k)
{
    const uint64_t p0 = uint64_t(0xD2511F53u) * c.x;
    const uint64_t p1 = uint64_t(0xCD9E8D57u) * c.z;
    if constexpr (VECTOR_XOR3)
        c = make_uint4(__builtin_amdgcn_bitop3_b32(uint32_t(p1 >> 32), c.y, k.x, 0x96), uint32_t(p1),
                       __builtin_amdgcn_bitop3_b32(uint32_t(p0 >> 32), c.w, k.y, 0x96), uint32_t(p0));
    else
        c = make_uint4(uint32_t(p1 >> 32) ^ c.y ^ k.x, uint32_t(p1), uint32_t(p0 >> 32) ^ c.w ^ k.y, uint32_t(p0));
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
}

// Every call site keeps the lane-varying index in counter word 1 (c.y) and everything else wave-uniform.
// The variation then reaches the multiplied words only gradually: round 1 is entirely uniform, rounds 2
// and 3 have one uniform multiply each.  Those rounds are written with plain xors so that hipcc keeps
// their uniform halves on the scalar unit (s_mul_hi_u32 / s_mul_i32 / s_xor); from round 4 on everything
// varies and the xors are v_bitop3.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k)
{
#ifdef ISINGMC_TIMING_ONLY_CHEAP_RNG // diagnostic build: what everything except Philox costs (results are wrong)
    return make_uint4(c.x ^ k.x ^ c.y, c.y * 3u + c.w, c.z ^ k.y ^ (c.y >> 3), c.w + c.y);
#endif
    philox_round<false>(c, k);
    philox_round<false>(c, k);
    philox_round<false>(c, k);
#pragma unroll
    for (int r = 3; r < 10; r++) philox_round<true>(c, k);
    return c;
}

// ---- calls that differ only in counter word 3 (and whose word 1 is the only lane-varying input) ----------------------------
// Rounds 1-3 of such calls, spelled out (M0, M1 the multipliers; W0, W1 the key increments; c = (c0, c1, c2, c3), key (kx, ky)):
//   A = M0 c0, B = M1 c2                                  -> (B.hi ^ c1 ^ kx, B.lo, A.hi ^ c3 ^ ky, A.lo)
//   P = M0 (c1 ^ E0), Q = M1 (A.hi ^ c3 ^ ky)             -> (Q.hi ^ B.lo ^ kx', Q.lo, P.hi ^ E1, P.lo)        E0 = B.hi ^ kx, E1 = A.lo ^ ky'
//   U = M0 (Q.hi ^ B.lo ^ kx'), V = M1 (P.hi ^ E1)        -> (V.hi ^ S1, V.lo, P.lo ^ S2, S3)       S1 = Q.lo ^ kx'', S2 = U.hi ^ ky'', S3 = U.lo
// P and V are the only lane-varying products and do not depend on c3: a thread that draws several calls computes them once
// (philox_shared).  Everything else is wave-uniform AND the same for every wave of a launch: per (key, c0, c2, c3) the three words
// S1, S2, S3 plus E1 -- philox_uniform_words, run once per launch by a table kernel instead of once per wave on the scalar unit
// (12 scalar instructions per call; the one-degree packed kernel draws 8 calls per wave and is bound by instruction issue).
constexpr uint32_t PHILOX_M0 = 0xD2511F53u, PHILOX_M1 = 0xCD9E8D57u, PHILOX_W0 = 0x9E3779B9u, PHILOX_W1 = 0xBB67AE85u;

struct PhiloxUniform {
    uint32_t s1, s2, s3, e1;
};

__host__ __device__ inline PhiloxUniform philox_uniform_words(uint32_t c0, uint32_t c2, uint32_t c3, uint2 key)
{
    const uint64_t A = uint64_t(PHILOX_M0) * c0, B = uint64_t(PHILOX_M1) * c2;
    const uint64_t Q = uint64_t(PHILOX_M1) * (uint32_t(A >> 32) ^ c3 ^ key.y);
    const uint64_t U = uint64_t(PHILOX_M0) * (uint32_t(Q >> 32) ^ uint32_t(B) ^ (key.x + PHILOX_W0));
    PhiloxUniform u;
    u.s1 = uint32_t(Q) ^ (key.x + 2u * PHILOX_W0);
    u.s2 = uint32_t(U >> 32) ^ (key.y + 2u * PHILOX_W1);
    u.s3 = uint32_t(U);
    u.e1 = uint32_t(A) ^ (key.y + PHILOX_W1);
    return u;
}

struct PhiloxShared {
    uint32_t p_lo, v_hi, v_lo;
};

// c2 is a compile-time constant at every call site (the domain word): B folds away
__device__ __forceinline__ PhiloxShared philox_shared(uint32_t c1, uint32_t c2, uint2 key, uint32_t e1)
{
    const uint32_t e0 = uint32_t((uint64_t(PHILOX_M1) * c2) >> 32) ^ key.x;
    const uint64_t P = uint64_t(PHILOX_M0) * (c1 ^ e0);
    const uint64_t V = uint64_t(PHILOX_M1) * (uint32_t(P >> 32) ^ e1);
    return PhiloxShared{uint32_t(P), uint32_t(V >> 32), uint32_t(V)};
}

// rounds 4-10 of the call whose uniform words are (s1, s2, s3)
__device__ __forceinline__ uint4 philox4x32_10_late(const PhiloxShared &sh, uint32_t s1, uint32_t s2, uint32_t s3, uint2 key)
{
    uint4 c = make_uint4(sh.v_hi ^ s1, sh.v_lo, sh.p_lo ^ s2, s3);
#ifdef ISINGMC_TIMING_ONLY_CHEAP_RNG
    return c;
#endif
    uint2 k = make_uint2(key.x + 3u * PHILOX_W0, key.y + 3u * PHILOX_W1);
#pragma unroll
    for (int r = 3; r < 10; r++) philox_round<true>(c, k);
    return c;
}

// The round keys of rounds 4-10 in VECTOR registers.  Measured on gfx950 (tools/ubench/valu_forms.hip): a VALU
// instruction with an SGPR source issues in 4.3 cycles per wave, the same instruction on VGPRs only in 2.5
// (v_xor, v_and, v_add, v_bitop3, shifts; everything else -- multiplies, v_and_or, v_alignbit, v_cndmask -- is
// 4.2 either way).  The xor3 of a Philox round has the (wave-uniform) round key as one source: 14 per call.
// Holding those 14 keys in VGPRs, written once per thread, takes 26 cycles off every call (150 -> 124).
// The v_mov is inline asm so that the compiler cannot fold the SGPR back into the uses.
struct PhiloxVKeys {
    uint32_t kx[7], ky[7];
};

__device__ __forceinline__ PhiloxVKeys philox_vkeys(uint2 k)
{
    PhiloxVKeys v;
#pragma unroll
    for (int r = 0; r < 7; r++) {
        const uint32_t x = k.x + uint32_t(r + 3) * 0x9E3779B9u, y = k.y + uint32_t(r + 3) * 0xBB67AE85u;
#ifdef ISINGMC_AB_SCALAR_KEYS // A/B build: the keys stay SGPR operands
        v.kx[r] = x;
        v.ky[r] = y;
#else
        asm volatile("v_mov_b32 %0, %1" : "=v"(v.kx[r]) : "s"(x));
        asm volatile("v_mov_b32 %0, %1" : "=v"(v.ky[r]) : "s"(y));
#endif
    }
    return v;
}

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k, const PhiloxVKeys &vk)
{
#ifdef ISINGMC_TIMING_ONLY_CHEAP_RNG
    return make_uint4(c.x ^ k.x ^ c.y, c.y * 3u + c.w, c.z ^ k.y ^ (c.y >> 3), c.w + c.y);
#endif
    philox_round<false>(c, k);
    philox_round<false>(c, k);
    philox_round<false>(c, k);
#ifdef ISINGMC_TIMING_ONLY_PHILOX_ROUNDS // diagnostic build (another generator: results differ): Philox4x32-R for R = 7 .. 10
    constexpr int late_rounds = ISINGMC_TIMING_ONLY_PHILOX_ROUNDS - 3;
#else
    constexpr int late_rounds = 7;
#endif
#pragma unroll
    for (int r = 0; r < late_rounds; r++) {
        const uint64_t p0 = uint64_t(0xD2511F53u) * c.x;
        const uint64_t p1 = uint64_t(0xCD9E8D57u) * c.z;
        c = make_uint4(__builtin_amdgcn_bitop3_b32(uint32_t(p1 >> 32), c.y, vk.kx[r], 0x96), uint32_t(p1),
                       __builtin_amdgcn_bitop3_b32(uint32_t(p0 >> 32), c.w, vk.ky[r], 0x96), uint32_t(p0));
    }
    return c;
}

// counter word 2: (t >> 32) in the top 16 bits, colour in bits 8..15, call index in bits 0..7
__device__ __forceinline__ uint32_t ctr2(uint64_t t, uint32_t colour, uint32_t call)
{
    return (uint32_t((t >> 32) & 0xFFFFu) << 16) | ((colour & 0xFFu) << 8) | (call & 0xFFu);
}

} // namespace isingmc
