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
// two more words for launches with ONE beta (every threshold word 0 or ~0): bit row * N_PLANES + plane = that word's bit
constexpr uint32_t PK_TAB_SEL = PK_TAB_LO + PK_MAX_DEG * 32;
constexpr uint32_t PK_TAB_WORDS = PK_TAB_SEL + 2;

constexpr uint32_t PK_NO_NBR = 0xFFFFFFFFu;

// Block headers of the ELL table: block B = positions [64 B, 64 B + 64) -- exactly what the 64 lanes of a wave
// touch for one word of their quads.  Where a slot's 64 entries are one translation (same offset to the own
// position, same coupling sign: the interior of any lattice-like graph) or all unused, the header replaces
// them: the wave reads 8 bytes through the scalar unit instead of 256 from the table.
constexpr uint32_t PK_HDR_MIXED = 0, PK_HDR_UNIFORM = 1, PK_HDR_UNUSED = 2; // .x bits 0-1; .x bit 31 = J > 0; .y = offset
constexpr uint32_t PK_HDR_UNIFORM_BUT_ONE = 3;                              // (one-degree kernels' shift headers only)

struct PkGraphDev {
    const uint2 *ell_hdr;       // [n_pos / 64][PK_MAX_DEG]
    const uint32_t *nbr_ell;    // [PK_MAX_DEG][n_pos]: neighbour position | (J > 0) << 31, or PK_NO_NBR
    const uint32_t *site;       // original site per position, PAD_SITE on padding
    const uint32_t *class_base; // n_colours + 1, multiples of 256
    uint32_t n_colours;
    uint32_t n_pos;             // multiple of 256
};

// Block headers of the one-degree kernels (packed_uni_kernels.hpp), [n_pos / 64][PK_MAX_DEG] each:
//   shift[B][i] = {PK_HDR_UNIFORM, 4 d} when slot i of the 64 positions of block B is one translation p -> p + d
//                 (whatever the signs; the shift in BYTES: the table-free instantiation adds it to a byte offset directly);
//                 {PK_HDR_UNIFORM_BUT_ONE | lane << 2 | (e & 0xFFFFFF) << 8, 4 d} when it is that translation for every lane but
//                 one, whose neighbour is p + d + e (-2^23 <= e < 2^23): where a row of a periodic lattice wraps around inside
//                 the block -- on BASELINE c5's 256^3 lattice two of the 24 (block, slot) pairs of EVERY wave, which each cost a
//                 dependent table read before the gathers could be issued (round 4: +8 %); honoured by the table-free
//                 instantiation only -- a launch that reads table entries anyway takes these from the table too;
//                 else {PK_HDR_MIXED, 0}: addresses from nbr_ell;
//   sign[B][i]  = bit l set <=> the slot-i bond of position 64 B + l has J > 0 (only read when the signs differ)
struct PkUniHeaders {
    const uint2 *shift;
    const uint2 *sign;
    uint32_t negmask; // one sign for every bond: 0 (J > 0) or ~0 (J < 0)
};

// One-degree kernel: every position in [class_begin, class_end) is a real site of degree `degree` (3..6);
// class_end - class_begin is a multiple of 256.  mixed_signs: couplings of both signs (else H.negmask holds the one sign).
// uniform_beta: every word of the threshold tables is 0 or ~0 (one beta for all replicas of the launch).
// n_groups: grid.y (the launch covers positions [class_begin, class_end) of that many replica groups)
hipError_t pk_uni_launch_sweep(int degree, bool uniform_beta, bool mixed_signs, uint32_t n_groups, hipStream_t stream, uint32_t *state,
                               const PkGraphDev &G, const PkUniHeaders &H, uint32_t class_begin, uint32_t class_end, uint64_t t,
                               const uint2 *group_keys, const uint32_t *tabs, uint32_t tab_stride, const uint32_t *philox_tab,
                               bool needs_table = true);
// needs_table = false: no (block, slot) header of [class_begin, class_end) is PK_HDR_MIXED (the host has checked): the instantiation
// without table reads
// philox_tab: the wave-uniform halves of timestep t's Philox calls for the launch's groups, pk_uni_philox_table_words() words per group,
// written by pk_uni_launch_philox_table for timesteps t0 .. t0 + n_steps - 1 as out[(k n_groups + g) words + ...] from the groups' keys
uint32_t pk_uni_philox_table_words();
hipError_t pk_uni_launch_philox_table(hipStream_t stream, uint32_t *out, const uint2 *group_keys, uint32_t n_groups, uint64_t t0, uint32_t n_steps);

} // namespace isingmc
