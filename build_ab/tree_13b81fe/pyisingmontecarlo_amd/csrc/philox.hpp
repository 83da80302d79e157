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
