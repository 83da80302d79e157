// Shared declarations of the replica-packed path: layout constants and the device view of the graph
// (packed_kernels.hpp: every graph; packed_uni_kernels.hpp: graphs of one degree and one coupling sign).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lattice_kernels.hpp"

namespace isingmc {

constexpr int PK_MAX_DEG = 6;
constexpr uint32_t DOM_PK_SWEEP = 0x504B5357u; // "PKSW"
constexpr uint32_t DOM_PK_INIT = 0x504B494Eu;  // "PKIN"
constexpr uint32_t PK_TAB_ALL = 0, PK_TAB_TBW = PK_MAX_DEG, PK_TAB_LO = PK_MAX_DEG + PK_MAX_DEG * N_PLANES;
constexpr uint32_t PK_TAB_WORDS = PK_TAB_LO + PK_MAX_DEG * 32;

constexpr uint32_t PK_NO_NBR = 0xFFFFFFFFu;

// Block headers of the ELL table: block B = positions [64 B, 64 B + 64) -- exactly what the 64 lanes of a wave
// touch for one word of their quads.  Where a slot's 64 entries are one translation (same offset to the own
// position, same coupling sign: the interior of any lattice-like graph) or all unused, the header replaces
// them: the wave reads 8 bytes through the scalar unit instead of 256 from the table.
constexpr uint32_t PK_HDR_MIXED = 0, PK_HDR_UNIFORM = 1, PK_HDR_UNUSED = 2; // .x bits 0-1; .x bit 31 = J > 0; .y = offset

struct PkGraphDev {
    const uint2 *ell_hdr;       // [n_pos / 64][PK_MAX_DEG]
    const uint32_t *nbr_ell;    // [PK_MAX_DEG][n_pos]: neighbour position | (J > 0) << 31, or PK_NO_NBR
    const uint32_t *site;       // original site per position, PAD_SITE on padding
    const uint32_t *class_base; // n_colours + 1, multiples of 256
    uint32_t n_colours;
    uint32_t n_pos;             // multiple of 256
};

// Block headers of the one-degree kernels (packed_uni_kernels.hpp), [n_pos / 64][PK_MAX_DEG] each:
//   shift[B][i] = {PK_HDR_UNIFORM, d} when slot i of the 64 positions of block B is one translation p -> p + d
//                 (whatever the signs), else {PK_HDR_MIXED, 0}: addresses from nbr_ell;
//   sign[B][i]  = bit l set <=> the slot-i bond of position 64 B + l has J > 0 (only read when the signs differ)
struct PkUniHeaders {
    const uint2 *shift;
    const uint2 *sign;
    uint32_t negmask; // one sign for every bond: 0 (J > 0) or ~0 (J < 0)
};

// One-degree kernel: every position in [class_begin, class_end) is a real site of degree `degree` (3..6);
// class_end - class_begin is a multiple of 256.  mixed_signs: couplings of both signs (else H.negmask holds the one sign).
// uniform_beta: every word of the threshold tables is 0 or ~0 (one beta for all replicas of the launch).
hipError_t pk_uni_launch_sweep(int degree, bool uniform_beta, bool mixed_signs, dim3 grid, hipStream_t stream, uint32_t *state,
                               const PkGraphDev &G, const PkUniHeaders &H, uint32_t class_begin, uint32_t class_end, uint64_t t,
                               const uint2 *group_keys, const uint32_t *tabs, uint32_t tab_stride);

} // namespace isingmc
