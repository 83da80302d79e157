// Host-side logic of libisingmc.so that needs no device: seeds, schedules, lattice recogniser,
// adjacency + greedy colouring.  See include/isingmc.h for the reference lines each part replaces.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace isingmc {

// xoshiro256++ seeded by SplitMix64: rand 0.8 SmallRng::seed_from_u64 on 64-bit targets
// (published algorithm; what lattice.rs:85-90 instantiates).
struct SmallRng {
    uint64_t s[4];
    explicit SmallRng(uint64_t seed);
    uint64_t next_u64();
};

std::vector<uint64_t> make_seeds(bool has_seed, uint64_t seed_gen, size_t n);

// lattice.rs:320-334 + 358-365; returns "" or an error message
std::string expand_schedule(const uint64_t *stop_t, const double *stop_beta, size_t n_stops,
                            size_t timesteps, bool compat_constant_beta, double *betas_out);

struct Lattice2D {
    bool ok = false;
    int W = 0, H = 0;
    double jabs = 0.0;                  // |J| of the horizontal bonds (of every bond when isotropic)
    double jabs_y = 0.0;                // |J| of the vertical bonds
    bool uniform_sign = true;
    bool jpos_uniform = false;          // sign when uniform: true = J > 0 (antiferromagnetic)
    bool open_x = false, open_y = false; // no bonds between columns W-1 and 0 / rows H-1 and 0 (all of them absent)
    std::vector<uint8_t> jright, jdown; // per site: 1 if that bond has J > 0 (empty when uniform)
};

Lattice2D recognise_lattice2d(const uint64_t *ea, const uint64_t *eb, const double *ej,
                              size_t n_edges, size_t nvars);

// adjacency in edge-list order (self-loops dropped, their J summed into self_energy)
struct Adjacency {
    std::vector<uint64_t> ptr; // nvars + 1
    std::vector<uint32_t> nbr;
    std::vector<double> w;
    double self_energy = 0.0;
};

Adjacency build_adjacency(const uint64_t *ea, const uint64_t *eb, const double *ej, size_t n_edges,
                          size_t nvars);

struct Colouring {
    std::vector<uint32_t> colour;     // per site
    uint32_t n_colours = 0;
    std::vector<uint64_t> class_base; // n_colours + 1, packed positions, classes padded to 256
    std::vector<uint64_t> pos;        // site -> packed position
    uint64_t n_pos = 0;
};

Colouring greedy_colouring(const Adjacency &A, size_t nvars);

// One parallel-tempering exchange round on the beta ladder.  Rung i (beta_i) holds
// replica slot perm[i]; round parity picks the pairs (i, i+1); swap iff
// u < exp((beta_i - beta_j)(E_i - E_j)), u from Philox keyed by (seed, round, i).  Swaps exchange the
// perm entries (temperatures move, configurations stay).  Returns the number of accepted swaps.
uint64_t pt_swap_round(uint64_t seed, uint64_t round, size_t n_rungs, const double *betas,
                       const double *slot_energy, uint32_t *perm);

// ---- replica-packed real-coupling path: host halves of the spec ---------------------------
// Scales: F_i = |h_i| + sum_e |J_e|, Fmax = max F_i, med = the lower median nonzero |coupling or bias|;
// k = ilogb(min(Fmax, 64 med)) + 1 - 30 is the graph's quantum; site i quantises what it sees at k_i = max(k, ilogb(F_i) + 1 - 30)
// (dshift[i] = min(k_i - k, 31); > 0: a HEAVY site -- one pinning bias, one enormous bond): jq in ADJACENCY order (A.w's, each
// entry as seen from its row's site), hq per site.  eligible: degree <= 31, Fmax > 0, and every heavy site dominated by one
// term (4 max term >= 3 F_i: no cancellation among its large terms can make the coarser quantum matter).
// Energy levels (the ORIGINAL couplings, not the dynamics' rounded ones): k_energy = ilogb(Fmax) + 2 - 30,
// x ~ hi 2^k_energy + lo 2^(k_energy - 24); jhi / jlo in adjacency order (one value per bond), hhi / hlo per site.
constexpr int RJ_ENERGY_LO_BITS = 24;
struct RjQuant {
    bool eligible = false, heavy = false;
    int k = 0, k_energy = 0;
    uint32_t max_degree = 0;
    std::vector<int32_t> jq, hq, jhi, jlo, hhi, hlo;
    std::vector<uint8_t> dshift;
};
RjQuant rj_quantise(const Adjacency &A, size_t nvars, const double *biases);
// acceptance scale of beta: accept iff max(X >> shift, 0) <= (Lambda_q(u) * mant) >> 32
void rj_beta(double beta, int k, uint32_t *shift_out, uint32_t *mant_out);
// LT[i], i = 0 .. 2048: log2(1 + i/2048) in Q24, centred for linear interpolation (see oracle/ising_oracle.c engine E)
void rj_log_table(uint32_t *out);

// packed checkerboard planes of one replica -> W*H bytes in site order (16 bytes per SSE2 store)
void unpack_lattice(uint32_t W, uint32_t H, const uint32_t *words, uint8_t *spins);

} // namespace isingmc
