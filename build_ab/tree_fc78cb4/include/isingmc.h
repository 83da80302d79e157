/*
 * isingmc.h -- C ABI of libisingmc.so, the MI355X (gfx950) classical Ising Metropolis engine.
 *
 * This is the drop-in boundary for the hot path of Renmusxd/PyIsingMonteCarlo: the reference's
 * pyo3 shell (src/lattice.rs, src/classicising.rs) drives one `qmc::classical::graph::GraphState`
 * per experiment from a rayon loop; a maintainer replaces that loop with the calls below
 * (INTEGRATION.md shows the `extern "C"` block).  Each entry point names the reference
 * interface it replaces.  Plain pointers and sizes only; the caller owns every host buffer; the
 * handles own all device memory.  Every function returns ISINGMC_OK or an error code and never
 * aborts the process (the reference aborts on engine errors: lattice.rs:206 + Cargo.toml:14);
 * isingmc_last_error() returns the message of the calling thread's last failure.
 *
 * There is NO CPU fallback: without a usable HIP device every device entry point fails with
 * ISINGMC_ERR_NO_DEVICE.  The isingmc_host_* helpers are pure host code and need no device.
 *
 * Hamiltonian: E = sum_edges J_ab s_a s_b - sum_i h_i s_i, s = +1 for True (README.md:45-46;
 * bias sign [UNVERIFIED]).
 */
#ifndef ISINGMC_H
#define ISINGMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISINGMC_ABI_VERSION 4

enum {
    ISINGMC_OK = 0,
    ISINGMC_ERR_INVALID = 1,   /* bad argument: maps to Python ValueError */
    ISINGMC_ERR_NO_DEVICE = 2, /* no HIP device / bad ordinal: RuntimeError */
    ISINGMC_ERR_HIP = 3,       /* a HIP runtime call failed: RuntimeError */
    ISINGMC_ERR_ALLOC = 4      /* host or device allocation failed: MemoryError */
};

/* graph kinds reported by isingmc_graph_info */
enum {
    ISINGMC_KIND_GENERAL = 0,  /* greedy-coloured CSR path, any edge list */
    ISINGMC_KIND_LATTICE2D = 1 /* W x H square lattice, uniform |J|, periodic or open, optional uniform field: checkerboard path */
};

/* isingmc_graph_create flags */
#define ISINGMC_FLAG_FORCE_GENERAL 1u /* skip the lattice recogniser (BASELINE config c5) */
/* Fix the kernel family from the GRAPH alone, never from the number of experiments: experiment k of a call then depends on
 * seed k only (lattice.rs:83-91, 198), i.e. run_monte_carlo(beta, T, R)[k] is the same for every R > k.  Without the flag the
 * faster family for the given R is chosen (small R on the f64 CSR kernels, large R replica-packed), and the two families are
 * different Markov chains for the same Hamiltonian. */
#define ISINGMC_FLAG_STABLE_PATH 2u

typedef struct isingmc_graph isingmc_graph;   /* edges + biases: the per-experiment adjacency that
                                                 GraphState::new builds (lattice.rs:199), built once */
typedef struct isingmc_states isingmc_states; /* R replicas = R x GraphState<SmallRng> on one device */

typedef struct {
    int32_t kind;        /* ISINGMC_KIND_* */
    int32_t device;      /* HIP device ordinal */
    uint64_t nvars;      /* max index + 1 (lattice.rs:51-55) */
    uint64_t n_edges;
    int32_t width;       /* LATTICE2D: W (columns), else 0 */
    int32_t height;      /* LATTICE2D: H (rows), else 0 */
    double jabs;         /* LATTICE2D: the common |J| */
    int32_t uniform_sign;/* LATTICE2D: 1 if every bond has the same sign */
    uint32_t n_colours;  /* independent sets per timestep (2 on the lattice path) */
    uint64_t state_words;/* 32-bit words of packed spin state per replica */
    int32_t fast_path;   /* LATTICE2D: 0 = periodic, no field; 1 = uniform field (set_global_bias, lattice.rs:129-131;
                            ClassicIsing longitudinal, classicising.rs:69); 2 = open boundaries;
                            3 = anisotropic (|J| of the horizontal bonds != |J| of the vertical ones);
                            4 = open boundaries and a field */
    int32_t open_x, open_y; /* LATTICE2D: no bonds between columns W-1 and 0 / rows H-1 and 0 */
    double field;        /* LATTICE2D: the uniform bias h of E = sum J s s - h sum s (0 without) */
    double jabs_y;       /* LATTICE2D: |J| of the vertical bonds (jabs is then the horizontal bonds'; equal unless fast_path == 3) */
    int32_t field_signs; /* LATTICE2D: 1 when the biases are +-field from site to site (sign planes), field = |h| then */
    int32_t packed_degree; /* GENERAL: d in 3..6 when the replica-packed path may use its one-degree kernel
                              (every site has d neighbours, every coupling the same size), else 0 */
    int32_t real_slots;    /* GENERAL: 4, 7, 11, 15, 23 or 31 when the replica-packed REAL-COUPLING path applies (any f64 couplings,
                              lattice.rs:46-50, and any site biases, lattice.rs:104-131; degree <= real_slots), else 0 */
    int32_t real_quantum_log2; /* that path's DYNAMICS compute with couplings rounded to multiples of 2^real_quantum_log2
                              (2^-30 of the largest |h_i| + sum_e |J_e|, never more than 2^-24 of the median term; heavy sites:
                              2^-30 of their own |h_i| + sum_e |J_e|) */
    int32_t real_energy_log2;  /* ... and its ENERGIES are those of the original couplings to within 2^(real_energy_log2 - 25)
                              per term (two exact integer levels; = Fmax 2^-54) */
    int32_t real_heavy_sites;  /* sites that quantise at a coarser scale of their own (pinned by a large bias, ...) */
    int32_t stable_path;       /* 1 when created with ISINGMC_FLAG_STABLE_PATH */
    int32_t packed_but_one_headers; /* one-degree packed kernel: (64-position block, slot) pairs that are one translation for every
                              lane but one (a lattice row wrapping around inside the block) -- served without a table read */
} isingmc_graph_info_t;

const char *isingmc_last_error(void);
int isingmc_abi_version(void);
/* The library recycles freed device blocks, pinned host blocks, streams and events between calls (a call of the reference's
 * API creates and drops its replicas; for small lattices hipMalloc / hipFree cost more than the timesteps).  This hands everything
 * that is idle back to the runtime; returns the number of bytes released.  (ISINGMC_NO_ALLOC_CACHE=1 disables the recycling.) */
size_t isingmc_release_cached_resources(void);
int isingmc_device_count(int *count);

/* ---- host-only helpers (no device) ------------------------------------------------------ */

/* lattice.rs:83-91 make_seeds: master SmallRng (seed_from_u64(seed_gen), or OS entropy when
 * has_seed == 0) -> one u64 per experiment. */
int isingmc_host_make_seeds(int has_seed, uint64_t seed_gen, size_t n, uint64_t *seeds_out);

/* lattice.rs:320-334 + 358-365 (and 406-420 + 445-451): sort the (t, beta) stops, default
 * [(0,1),(T,1)], pad to [0,T], and expand to one beta per timestep by linear interpolation.
 * compat_constant_beta != 0 reproduces the reference's behaviour (the interpolation index is a
 * captured constant, so beta is the last stop's beta for the whole run). */
int isingmc_host_expand_schedule(const uint64_t *stop_t, const double *stop_beta, size_t n_stops,
                                 size_t timesteps, int compat_constant_beta, double *betas_out);

/* Recogniser: is this edge list a W x H square lattice with ids y*W+x, every bond present once,
 * one |J| per direction, periodic or open (ALL wrap-around bonds of a direction absent) in each direction?
 * *is_lattice = 0 when not (then the general path is used), else 1 + 2 (open in x) + 4 (open in y)
 * + 8 (|J| of the horizontal bonds, returned in *jabs, differs from the vertical bonds'). */
int isingmc_host_recognise_lattice2d(const uint64_t *edge_a, const uint64_t *edge_b,
                                     const double *edge_j, size_t n_edges, size_t nvars,
                                     int *is_lattice, int *width, int *height, double *jabs,
                                     int *uniform_sign);

/* Greedy colouring used by the general path (sites in index order, smallest free colour). */
int isingmc_host_colour_graph(const uint64_t *edge_a, const uint64_t *edge_b, size_t n_edges,
                              size_t nvars, uint32_t *colours_out, uint32_t *n_colours_out);

/* One exchange round of the classical parallel-tempering ladder -- the classical counterpart of
 * TemperingContainer::parallel_tempering_step driven from tempering.rs:191-194 (quantum in the
 * reference).  Rung i has inverse temperature betas[i] and currently holds replica slot perm[i];
 * slot_energy[s] is that slot's energy (all-gathered across ranks by the caller).  Pairs (i, i+1)
 * with i of the round's parity are swapped with probability min(1, exp((b_i-b_j)(E_i-E_j))) using
 * a Philox stream keyed by (seed, round, i): every rank computes the same decisions from the same
 * inputs.  perm is updated in place; *swaps_out receives the number of accepted swaps. */
int isingmc_host_pt_swap_round(uint64_t seed, uint64_t round, size_t n_rungs, const double *betas,
                               const double *slot_energy, uint32_t *perm, uint64_t *swaps_out);

/* Host halves of the replica-packed REAL-COUPLING path -- what the device kernels are fed with, exposed
 * so that they can be checked without a GPU.  That path serves edge lists with couplings of several sizes
 * (lattice.rs:46-50 takes any f64) and arbitrary site biases (set_individual_bias / set_global_bias, lattice.rs:104-131)
 * on graphs of degree <= 31.  Scales: F_i = |h_i| + sum_e |J_e|, Fmax = max_i F_i, med = the lower median nonzero
 * |coupling or bias|; the graph's quantum is 2^k, k = ilogb(min(Fmax, 64 med)) + 1 - 30; site i quantises what it sees in units
 * of 2^(k + d_i), d_i = max(0, ilogb(F_i) + 1 - 30 - k) capped at 31 (d_i > 0: a heavy site, e.g. one pinned by a large bias).
 * jq_out: TWO values per input edge -- the bond as seen from edge_a[e] and from edge_b[e] (0, 0 for self-loops); hq_out and
 * dshift_out: one per site; *eligible_out: degree <= 31, Fmax > 0 and every heavy site dominated by one term
 * (4 max(|h_i|, max_e |J_e|) >= 3 F_i) -- else the f64 CSR path is used. */
int isingmc_host_rj_quantise(const uint64_t *edge_a, const uint64_t *edge_b, const double *edge_j, size_t n_edges,
                             size_t nvars, const double *biases, int32_t *jq_out, int32_t *hq_out, uint8_t *dshift_out,
                             int *k_out, int *eligible_out);
/* the energy of that path is the energy of the ORIGINAL couplings in two exact integer levels:
 * x ~ hi 2^kE + lo 2^(kE - 24), kE = ilogb(Fmax) + 2 - 30, |x - (hi 2^kE + lo 2^(kE-24))| <= Fmax 2^-54;
 * jhi_out / jlo_out per input edge (0 for self-loops), hhi_out / hlo_out per site */
int isingmc_host_rj_energy_levels(const uint64_t *edge_a, const uint64_t *edge_b, const double *edge_j, size_t n_edges,
                                  size_t nvars, const double *biases, int32_t *jhi_out, int32_t *jlo_out, int32_t *hhi_out,
                                  int32_t *hlo_out, int *k_energy_out);
/* acceptance scale of one inverse temperature: a flip of site i with half energy change X (units of 2^(k + d_i)) is accepted iff
 * max(X >> (*shift_out - m), 0) <= ((Lambda_q(u) * *mant_out) >> 32) >> (d_i - m), m = min(*shift_out, d_i), Lambda_q(u) = 32 - log2(u) in Q24 for the 32-bit uniform u */
int isingmc_host_rj_beta(double beta, int k, uint32_t *shift_out, uint32_t *mant_out);
/* the 2049 entries of the log2(1 + i/2048) table behind Lambda_q (Q24, centred for linear interpolation) */
int isingmc_host_rj_log_table(uint32_t *table_out);

/* ---- graph: replaces the adjacency half of GraphState::new (lattice.rs:199, classicising.rs:73)
 * edges as three parallel arrays (the Vec<((usize,usize),f64)> of lattice.rs:47); biases NULL
 * (all zero) or nvars doubles (lattice.rs:186-189).  device = HIP ordinal. */
int isingmc_graph_create(const uint64_t *edge_a, const uint64_t *edge_b, const double *edge_j,
                         size_t n_edges, size_t nvars, const double *biases, int device,
                         unsigned flags, isingmc_graph **graph_out);
int isingmc_graph_info(const isingmc_graph *graph, isingmc_graph_info_t *info_out);
void isingmc_graph_destroy(isingmc_graph *graph);

/* ---- states: replaces R x { SmallRng::seed_from_u64(seed); GraphState::new(..., rng);
 * set_state(initial) } (lattice.rs:198-203) and GraphState::new_with_state_and_rng
 * (classicising.rs:71).  seeds[r] keys replica r's Philox stream (results do not depend on which
 * device or in which batch a replica runs).  initial_state: NULL (random start) or nvars bytes
 * (nonzero = True) copied into every replica.  The graph must outlive the states. */
int isingmc_states_create(isingmc_graph *graph, size_t n_replicas, const uint64_t *seeds,
                          const uint8_t *initial_state, isingmc_states **states_out);
/* One shard of the same fan-out: experiments [first, first + count) of the n_total whose seeds are
 * all_seeds[n_total] (the zip of lattice.rs:192-197 cut into contiguous blocks, one per GPU).  Every
 * choice that shapes a trajectory is made from the GLOBAL experiment index and count, so the union of
 * the shards' results equals one unsharded call whatever the cut.  isingmc_states_create(g, n, seeds)
 * is the shard [0, n) of n. */
int isingmc_states_create_range(isingmc_graph *graph, size_t n_total, const uint64_t *all_seeds,
                                size_t first, size_t count, const uint8_t *initial_state,
                                isingmc_states **states_out);
/* ClassicIsing.add_graph (classicising.rs:62-79): append one replica. */
int isingmc_states_append(isingmc_states *states, uint64_t seed, const uint8_t *initial_state);
/* GraphState::set_state (lattice.rs:202) on one replica. */
int isingmc_states_set_state(isingmc_states *states, size_t replica, const uint8_t *state);
size_t isingmc_states_count(const isingmc_states *states);
void isingmc_states_destroy(isingmc_states *states);

/* Per-replica inverse temperatures (parallel-tempering ladder; shaped after
 * LatticeTempering.add_graph(beta), tempering.rs:70-113).  NULL clears them.  While set, the
 * betas argument of isingmc_do_time_steps is ignored. */
int isingmc_states_set_betas(isingmc_states *states, const double *beta_per_replica);

/* replaces `for _ in 0..timesteps { gs.do_time_step(beta, None, None, None, only_basic) }`
 * (lattice.rs:204-207, 271-280, 358-368, 445-455; classicising.rs:97-109) for all replicas.
 * One timestep = one full sweep = nvars single-spin Metropolis attempts per replica.
 * Timestep k uses beta = betas[k * beta_stride] (stride 0: constant beta).
 * energies_per_step: NULL, or double[R][timesteps] receiving get_energy() after every timestep
 * (lattice.rs:454).  Blocking: returns after the device has finished. */
int isingmc_do_time_steps(isingmc_states *states, size_t timesteps, const double *betas,
                          size_t beta_stride, double *energies_per_step);
/* Same work, additionally reporting the device time of the sweep kernels (HIP events recorded on
 * the engine's stream around the launches) -- the measurement hook of bench.py. */
int isingmc_do_time_steps_timed(isingmc_states *states, size_t timesteps, const double *betas,
                                size_t beta_stride, float *device_ms_out);

/* GraphState::get_energy (lattice.rs:208, 284, 370; classicising.rs:171): double[R]. */
int isingmc_get_energies(isingmc_states *states, double *energies_out);
/* sum_i s_i per replica: int64[R] (the build's own observable for <|M|> parity). */
int isingmc_get_magnetisations(isingmc_states *states, int64_t *mags_out);
/* GraphState::get_state / state_ref (lattice.rs:209-211, 281-283): replica r's nvars spins as
 * bytes (1 = True) at states_out + r * replica_stride_bytes (stride >= nvars; lets the caller
 * write straight into a bool[R,S,N] array). */
int isingmc_get_states(isingmc_states *states, uint8_t *states_out, size_t replica_stride_bytes);
/* Raw packed device words of every replica (layout: DESIGN.md S2), uint32[R][state_words]. */
int isingmc_get_packed_states(isingmc_states *states, uint32_t *words_out);
/* Absolute timestep counter of the replicas (Philox counter word; persists across calls). */
uint64_t isingmc_states_timestep(const isingmc_states *states);
/* Sets that counter (t < 2^48): with the seeds, isingmc_states_set_state and this, a run that was stopped after t timesteps
 * resumes on exactly the trajectory it would have followed (the reference has no equivalent: its rng state is not exposed). */
int isingmc_states_set_timestep(isingmc_states *states, uint64_t t);

/* replaces the whole sampling loop of lattice.rs:271-287 / classicising.rs:144-173:
 *   thermalization x do_time_step(beta);  n_samples x { sampling_freq x do_time_step(beta);
 *   states[r][k][:] = state_ref();  energies[r][k] = get_energy() }
 * energies_out: double[R][n_samples]; states_out: bytes [R][n_samples][nvars] (the bool[R,S,N] array).
 * beta is ignored while per-replica betas are set.  Sweeps, sample copies and measurements are
 * enqueued back to back; the host waits once per chunk of samples. */
int isingmc_run_sampling(isingmc_states *states, double beta, size_t thermalization, size_t sampling_freq,
                         size_t n_samples, double *energies_out, uint8_t *states_out);

/* ---- on-stream parallel tempering (periodic field-free lattices; replica-packed real-coupling containers) ----
 * The classical counterpart of the loop in tempering.rs:177-194 { timesteps; parallel_tempering_step }
 * with NO host synchronisation inside it: sweeps, the energy measurement, the exchange decisions
 * (same arithmetic as isingmc_host_pt_swap_round) and the relabelling of the slots' betas are all
 * enqueued on the engine's HIP stream.  Between isingmc_pt_measure and isingmc_pt_swap a multi-GPU
 * caller all-gathers the `local` buffer of every rank into the `all` buffer ON THAT STREAM (RCCL:
 * ncclAllGather / torch.distributed.all_gather_into_tensor under torch.cuda.ExternalStream); with a
 * single rank isingmc_pt_measure fills `all` itself.
 *
 * attach: ladder_betas[n_rungs] in ladder order; this shard owns slots
 * [slot_offset, slot_offset + n_replicas) of n_rungs, every rank owning slots_per_rank slots (the last
 * ranks fewer); rung i starts on slot i; `seed` keys the exchange decisions. */
int isingmc_pt_attach(isingmc_states *states, const double *ladder_betas, size_t n_rungs, size_t slot_offset,
                      size_t slots_per_rank, size_t world_size, uint64_t seed);
/* would isingmc_pt_attach accept this container and geometry?  *ok_out = 1 / 0, no side effects (when 0,
 * isingmc_last_error() says why): the ranks of a sharded ladder agree on the answer BEFORE any of them attaches */
int isingmc_pt_can_attach(const isingmc_states *states, size_t n_rungs, size_t slot_offset, size_t slots_per_rank,
                          size_t world_size, int *ok_out);
/* release the ladder (synchronises); configurations and timestep stay, the per-replica betas are cleared */
int isingmc_pt_detach(isingmc_states *states);
/* device pointers: local = double[slots_per_rank] (send buffer), all = double[world_size*slots_per_rank] */
int isingmc_pt_buffers(isingmc_states *states, void **d_local_out, void **d_all_out, size_t *per_rank_out);
int isingmc_pt_time_steps(isingmc_states *states, size_t timesteps); /* enqueue only */
int isingmc_pt_measure(isingmc_states *states);                      /* enqueue only */
/* single rank: `timesteps` sweeps with an exchange round after every swap_every-th one, enqueued in one call (on
 * mid-size lattices one persistent launch whose strips exchange temperatures themselves; same decisions) */
int isingmc_pt_run(isingmc_states *states, size_t timesteps, size_t swap_every);
int isingmc_pt_swap(isingmc_states *states);                         /* enqueue only */
/* synchronises; perm_out = uint32[n_rungs] (rung -> slot), exchange rounds done, accepted swaps */
int isingmc_pt_state(isingmc_states *states, uint32_t *perm_out, uint64_t *round_out, uint64_t *swaps_out);
/* the engine's hipStream_t (for enqueuing the collective) and a host-side wait for it */
int isingmc_states_stream(isingmc_states *states, void **stream_out);
int isingmc_synchronize(isingmc_states *states);

/* ---- in-process ladder across several devices --------------------------------------------------------------
 * One host thread, one isingmc_states per device, each with the SAME ladder attached (isingmc_pt_attach with world_size =
 * n_shards, slot_offset = k * slots_per_rank): the group runs the exchange step of tempering.rs:191-194 between them without any
 * binding of the caller's to a collective library.  backend 0: RCCL (ncclCommInitAll + ncclAllGather on the engines' streams;
 * librccl.so is loaded with dlopen when the group is created, so single-GPU users need nothing) when every shard has a device of
 * its own and the library resolves, else event-ordered device copies (hipMemcpyPeerAsync); 1: RCCL or an error; 2: copies.
 * isingmc_pt_group_run enqueues { timesteps; measure; all-gather; swap } for the whole ladder; nothing waits on the host until
 * isingmc_pt_group_synchronize.  The shards stay the caller's (destroy the group first). */
typedef struct isingmc_pt_group isingmc_pt_group;
int isingmc_pt_group_create(isingmc_states **shards, size_t n_shards, int backend, isingmc_pt_group **group_out);
int isingmc_pt_group_backend(const isingmc_pt_group *group); /* 1 = RCCL, 2 = device copies */
int isingmc_pt_group_allgather(isingmc_pt_group *group);     /* enqueue only: local buffers -> every shard's all buffer */
int isingmc_pt_group_run(isingmc_pt_group *group, size_t timesteps, size_t swap_every); /* enqueue only */
int isingmc_pt_group_synchronize(isingmc_pt_group *group);
void isingmc_pt_group_destroy(isingmc_pt_group *group);

/* ---- measurement hook (bench.py; no reference counterpart) ------------------------------------------
 * Runs `timesteps` sweeps at `beta` like isingmc_do_time_steps and, beside them on a side stream, one
 * wave that stamps the shader-cycle counter against the 100 MHz constant counter for `probe_ms`
 * milliseconds: *ghz_out = the shader clock the chip holds under this kernel. */
int isingmc_debug_shader_clock(isingmc_states *states, size_t timesteps, double beta, double probe_ms,
                               double *ghz_out);

#ifdef __cplusplus
}
#endif
#endif /* ISINGMC_H */
