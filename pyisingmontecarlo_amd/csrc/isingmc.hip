// libisingmc.so: replica containers and every Monte-Carlo launch of the C ABI of include/isingmc.h (gfx950 only).
// Host orchestration only -- every Monte-Carlo operation runs in the kernels of lattice_kernels.hpp / general_kernels.hpp /
// packed_kernels.hpp (+ the strip, spread, multi-class, one-degree and real-coupling translation units).  There is no CPU fallback.
// (internal.hpp: how the host side is cut into translation units.)
#include "internal.hpp"

// ------------------------------------------------------------------------------------------------
// states
// ------------------------------------------------------------------------------------------------
static dim3 lat_grid(const isingmc_graph *g, uint32_t quads, size_t replicas)
{
    (void)g;
    return dim3((quads + 255) / 256, unsigned(replicas), 1);
}


static int lanes_reserve(isingmc_states *s, size_t n);
static int lanes_fork(isingmc_states *s, size_t n);
static int lanes_join(isingmc_states *s);

// replica-packed general path (defined further down)
static int choose_packed(const isingmc_states *s, size_t n_replicas);
static int pk_create(isingmc_states *s, const uint64_t *all_seeds, size_t first, size_t n, const uint8_t *initial_state);
static int pk_set_state(isingmc_states *s, size_t replica, const uint8_t *spins);
static int pk_set_betas(isingmc_states *s);
static int pk_append(isingmc_states *s, uint64_t seed, const uint8_t *initial_state);

// random start for replicas [first, first+count)
static int init_random(isingmc_states *s, size_t first, size_t count)
{
    const isingmc_graph *g = s->g;
    for (size_t r0 = first; r0 < first + count; r0 += MAX_GRID_Y) {
        const size_t n = std::min(MAX_GRID_Y, first + count - r0);
        if (g->kind == ISINGMC_KIND_LATTICE2D)
            hipLaunchKernelGGL(lat_init_kernel, lat_grid(g, 2 * g->geom.nquads, n), dim3(256), 0, s->stream,
                               s->d_state, g->geom, s->d_keys, uint32_t(r0));
        else
            hipLaunchKernelGGL(gen_init_kernel, dim3((g->gdev.n_words + 255) / 256, unsigned(n)), dim3(256), 0,
                               s->stream, s->d_state, g->gdev, s->d_keys, uint32_t(r0));
        HIP_TRY(hipGetLastError());
    }
    return ISINGMC_OK;
}

static int upload_state(isingmc_states *s, size_t first, size_t count, const uint8_t *spins)
{
    s->meas_fresh = false;
    std::vector<uint32_t> words(s->g->state_words);
    pack_state(s->g, spins, words.data());
    for (size_t r = first; r < first + count; r++)
        HIP_TRY(hipMemcpyAsync(s->d_state + r * s->g->state_words, words.data(), words.size() * sizeof(uint32_t),
                               hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return ISINGMC_OK;
}

static int reserve(isingmc_states *s, size_t cap)
{
    if (cap <= s->cap) return ISINGMC_OK;
    const isingmc_graph *g = s->g;
    uint32_t *d_state = nullptr;
    uint2 *d_keys = nullptr;
    struct Undo { // a failure below must not leak the new buffers
        uint32_t **a;
        uint2 **b;
        bool armed = true;
        ~Undo() { if (armed) { if (*a) (void)cached_free(*a); if (*b) (void)cached_free(*b); } }
    } undo{&d_state, &d_keys};
    TRY(dev_alloc(&d_state, cap * g->state_words));
    TRY(dev_alloc(&d_keys, cap));
    // enqueue-only calls (isingmc_pt_*, the sampling loop) may still be running on the engine's non-blocking stream, which the
    // null-stream copies below are NOT ordered against; and the old blocks go back to the cache at the end
    HIP_TRY(stream_quiesce(s->stream));
    for (auto st : s->lanes) HIP_TRY(stream_quiesce(st));
    if (s->copy_stream) HIP_TRY(stream_quiesce(s->copy_stream));
    if (s->R) {
        HIP_TRY(hipMemcpy(d_state, s->d_state, s->R * g->state_words * sizeof(uint32_t), hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(d_keys, s->d_keys, s->R * sizeof(uint2), hipMemcpyDeviceToDevice));
        HIP_TRY(hipDeviceSynchronize()); // device-to-device copies may still run when hipMemcpy returns; the old blocks are recycled below
    }
    undo.armed = false;
    for (void *p : {(void *)s->d_state, (void *)s->d_keys, (void *)s->d_thr, (void *)s->d_beta, (void *)s->d_meas,
                    (void *)s->d_pe, (void *)s->d_oe, (void *)s->d_pm, (void *)s->d_om})
        if (p) (void)cached_free(p);
    s->d_state = d_state;
    s->d_keys = d_keys;
    s->d_thr = nullptr; s->d_beta = nullptr; s->d_meas = nullptr;
    s->d_pe = nullptr; s->d_oe = nullptr; s->d_pm = nullptr; s->d_om = nullptr;
    TRY(dev_alloc(&s->d_thr, cap));
    TRY(dev_alloc(&s->d_beta, cap));
    if (g->kind == ISINGMC_KIND_LATTICE2D) {
        TRY(dev_alloc(&s->d_meas, 2 * cap));
        s->meas_zero = false;
    } else {
        s->n_partials = (g->gdev.n_pos + 255) / 256;
        TRY(dev_alloc(&s->d_pe, cap * s->n_partials));
        TRY(dev_alloc(&s->d_pm, cap * s->n_partials));
        TRY(dev_alloc(&s->d_oe, cap));
        TRY(dev_alloc(&s->d_om, cap));
    }
    s->cap = cap;
    return ISINGMC_OK;
}

static int add_replicas(isingmc_states *s, size_t count, const uint64_t *seeds, const uint8_t *initial_state)
{
    const size_t first = s->R;
    if (first + count > s->cap) TRY(reserve(s, std::max(first + count, s->cap + s->cap / 2)));
    std::vector<uint2> keys(count);
    for (size_t i = 0; i < count; i++) keys[i] = make_uint2(uint32_t(seeds[i]), uint32_t(seeds[i] >> 32));
    if (count) HIP_TRY(hipMemcpy(s->d_keys + first, keys.data(), count * sizeof(uint2), hipMemcpyHostToDevice));
    s->R = first + count;
    if (initial_state) TRY(upload_state(s, first, count, initial_state));
    else {
        TRY(init_random(s, first, count));
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    return ISINGMC_OK;
}

// Experiments [first, first + count) of n_total (one shard of the rayon fan-out of lattice.rs:192-197).  Everything that
// shapes a trajectory is decided from the GLOBAL experiment index and count -- the packed / per-replica choice, the
// 32-replica group a replica belongs to, the group's key and the replica's bit -- so that the results do not depend on
// how the experiments are cut into shards.  A shard that starts or ends inside a group simulates the whole group
// (the replicas of a group share Philox words and number their ties together).
extern "C" int isingmc_states_create_range(isingmc_graph *g, size_t n_total, const uint64_t *all_seeds, size_t first,
                                           size_t count, const uint8_t *initial_state, isingmc_states **states_out)
{
    if (!g || !states_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    *states_out = nullptr;
    if (n_total && !all_seeds) return fail(ISINGMC_ERR_INVALID, "seeds is NULL");
    if (first > n_total || count > n_total - first) return fail(ISINGMC_ERR_INVALID, "replica range out of bounds");
    TRY(use_device(g->device));
    auto s = std::make_unique<isingmc_states>();
    s->g = g;
    HIP_TRY(pooled_stream_create(&s->stream));
    HIP_TRY(pooled_event_create(&s->ev0, false));
    HIP_TRY(pooled_event_create(&s->ev1, false));
    TRY(lanes_reserve(s.get(), 2)); // created up front: the first multi-lane run must not pay for stream creation
    s->n_total = n_total;
    s->first = first;
    s->opt = Options::from_env();
    if (const int mode = count ? choose_packed(s.get(), n_total) : 0) {
        s->rj = mode == 2;
        TRY(pk_create(s.get(), all_seeds, first, count, initial_state));
    } else {
        TRY(reserve(s.get(), std::max<size_t>(count, 1)));
        TRY(add_replicas(s.get(), count, all_seeds + first, initial_state));
    }
    *states_out = s.release();
    return ISINGMC_OK;
}

extern "C" int isingmc_states_create(isingmc_graph *g, size_t n_replicas, const uint64_t *seeds,
                                     const uint8_t *initial_state, isingmc_states **states_out)
{
    return isingmc_states_create_range(g, n_replicas, seeds, 0, n_replicas, initial_state, states_out);
}

extern "C" int isingmc_states_append(isingmc_states *s, uint64_t seed, const uint8_t *initial_state)
{
    if (!s) return fail(ISINGMC_ERR_INVALID, "NULL states");
    if (s->has_betas) return fail(ISINGMC_ERR_INVALID, "clear the per-replica betas before appending replicas");
    TRY(use_device(s->g->device));
    if (s->packed) return pk_append(s, seed, initial_state);
    return add_replicas(s, 1, &seed, initial_state);
}

extern "C" int isingmc_states_set_state(isingmc_states *s, size_t replica, const uint8_t *state)
{
    if (!s || !state) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    if (replica >= s->R) return fail(ISINGMC_ERR_INVALID, "replica index out of range");
    TRY(use_device(s->g->device));
    if (s->packed) return pk_set_state(s, replica, state);
    return upload_state(s, replica, 1, state);
}

// kernel family: 0 checkerboard lattice kernels, 1 f64 CSR (one replica per word set), 2 replica-packed bit-sliced, 3 replica-packed real-coupling
extern "C" int isingmc_states_family(const isingmc_states *s, int *family_out)
{
    if (!s || !family_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    *family_out = s->g->kind == ISINGMC_KIND_LATTICE2D ? 0 : !s->packed ? 1 : s->rj ? 3 : 2;
    return ISINGMC_OK;
}

// ... and the family a container of n_experiments created NOW on this graph would take (the environment's switches as they are)
extern "C" int isingmc_graph_family_for(const isingmc_graph *g, size_t n_experiments, int *family_out)
{
    if (!g || !family_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    if (g->kind == ISINGMC_KIND_LATTICE2D) { *family_out = 0; return ISINGMC_OK; }
    isingmc_states probe;
    probe.g = const_cast<isingmc_graph *>(g);
    probe.opt = Options::from_env();
    const int mode = n_experiments ? choose_packed(&probe, n_experiments) : 0;
    probe.g = nullptr; // (the probe owns nothing: its destructor must not touch the device)
    *family_out = mode == 0 ? 1 : mode == 1 ? 2 : 3;
    return ISINGMC_OK;
}

extern "C" size_t isingmc_states_count(const isingmc_states *s) { return s ? s->R : 0; }

extern "C" uint64_t isingmc_states_timestep(const isingmc_states *s) { return s ? s->t : 0; }

extern "C" int isingmc_states_set_timestep(isingmc_states *s, uint64_t t)
{
    if (!s) return fail(ISINGMC_ERR_INVALID, "NULL states");
    // the counter words hold 48 bits of t (philox.hpp ctr2: t_lo in word 0, bits 32..47 beside the colour / call index)
    if (t >> 48) return fail(ISINGMC_ERR_INVALID, "timestep counter must be below 2^48");
    TRY(use_device(s->g->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->t = t;
    return ISINGMC_OK;
}

extern "C" void isingmc_states_destroy(isingmc_states *s) { delete s; }

extern "C" int isingmc_states_set_option(isingmc_states *s, const char *name, long value)
{
    if (!s || !name) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    const std::string n(name);
    for (const char *family : {"force_real", "disable_real", "force_packed", "disable_packed"})
        if (n.find(family) != std::string::npos && (n.size() == std::strlen(family) || n.size() == std::strlen(family) + 8))
            return fail(ISINGMC_ERR_INVALID, "the kernel family of a container is fixed when it is created (set ISINGMC_" + std::string(family) +
                                                 " in the environment before isingmc_states_create)");
    if (!s->opt.set(n, value)) return fail(ISINGMC_ERR_INVALID, "unknown option '" + n + "'");
    return ISINGMC_OK;
}

extern "C" int isingmc_states_set_betas(isingmc_states *s, const double *beta_per_replica)
{
    return set_betas(s, beta_per_replica, false);
}

// all_equal: the caller passes one beta R times (run_sampling) -- then a shard that cuts a replica group is fine
int set_betas(isingmc_states *s, const double *beta_per_replica, bool all_equal)
{
    if (!s) return fail(ISINGMC_ERR_INVALID, "NULL states");
    if (!beta_per_replica) {
        s->has_betas = false;
        s->betas.clear();
        return ISINGMC_OK;
    }
    for (size_t r = 0; r < s->R; r++)
        if (!std::isfinite(beta_per_replica[r])) return fail(ISINGMC_ERR_INVALID, "beta must be finite");
    TRY(use_device(s->g->device));
    if (!all_equal) { // one beta R times is as good as a uniform beta: any cut of a group is fine then
        all_equal = true;
        for (size_t r = 1; r < s->R; r++) all_equal &= beta_per_replica[r] == beta_per_replica[0];
    }
    if (s->packed && !s->rj && !all_equal && (s->pk_bit0 != 0 || ((s->first + s->R) % 32 != 0 && s->first + s->R != s->n_total)))
        // the replicas of a group number their ties together: a group's trajectory depends on all 32 betas, and this
        // shard only knows its own (the real-coupling path decides every replica on its own: any cut is fine there)
        return fail(ISINGMC_ERR_INVALID, "per-replica betas on a replica-packed shard: the shard must start and end on multiples of 32 experiments (or at the last experiment)");
    s->betas.assign(beta_per_replica, beta_per_replica + s->R);
    if (s->packed) {
        if (s->R) TRY(pk_set_betas(s));
        s->has_betas = true;
        return ISINGMC_OK;
    }
    if (s->g->kind == ISINGMC_KIND_LATTICE2D && s->g->mc_mode != MC_NONE) {
        std::vector<LatThrMC> thr(s->R);
        for (size_t r = 0; r < s->R; r++) thr[r] = lattice_thresholds_mc(s->g, s->betas[r]);
        HIP_TRY(stream_quiesce(s->stream)); // the old table may still be read by enqueued timesteps; its block is recycled
        if (s->d_thr_mc) HIP_TRY(cached_free(s->d_thr_mc));
        s->d_thr_mc = nullptr;
        TRY(dev_alloc(&s->d_thr_mc, s->cap));
        if (s->R) HIP_TRY(hipMemcpyAsync(s->d_thr_mc, thr.data(), s->R * sizeof(LatThrMC), hipMemcpyHostToDevice, s->stream));
    } else if (s->g->kind == ISINGMC_KIND_LATTICE2D) {
        std::vector<LatThr> thr(s->R);
        for (size_t r = 0; r < s->R; r++) thr[r] = lattice_thresholds(s->betas[r], s->g->jabs);
        if (s->R) HIP_TRY(hipMemcpyAsync(s->d_thr, thr.data(), s->R * sizeof(LatThr), hipMemcpyHostToDevice, s->stream));
    } else if (s->R) {
        HIP_TRY(hipMemcpyAsync(s->d_beta, s->betas.data(), s->R * sizeof(double), hipMemcpyHostToDevice, s->stream));
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->has_betas = true;
    return ISINGMC_OK;
}

// ------------------------------------------------------------------------------------------------
// replica-packed general path (packed_kernels.hpp): state = uint32[groups][n_pos], group g = replicas
// 32g .. 32g+31, keyed by the seed of its first replica
// ------------------------------------------------------------------------------------------------

// worth it from 16 replicas on and when the graph is too big for the LDS-resident per-replica kernel
// LDS-resident kernel for small graphs only: one workgroup walks a whole replica, ~13 us + 2.8..5 ns per site
// and timestep whatever the replica count, against ~5 us per colour class for the per-colour launches --
// measured crossover ~8 000 sites at 4 replicas, ~14 000 at 64 (profiles/r01_resident_threshold.txt); a
// 200 000-site graph ran 13x slower resident than streamed.
static bool gen_resident_fits(const isingmc_graph *g, size_t n_replicas)
{
    return g->state_words * sizeof(uint32_t) <= GEN_RESIDENT_MAX_BYTES && g->nvars <= (n_replicas < 16 ? 8000u : 12000u);
}

// Packed or per-replica?  The packed kernels launch once per colour class and timestep; the LDS-resident CSR kernel runs a whole
// call in one launch with one workgroup per replica, which wins on small graphs (they are bound by parallelism, and a word of 32
// replicas concentrates them on few compute units).  Measured (profiles/r03_real_small.txt, r03_few_replicas.txt,
// r03_packed_resident_experiment.txt; attempts/s of the packed path over the per-replica one):
//  * real-coupling path, graphs the resident CSR kernel takes: 8 000 sites 1.7x at 16 replicas, 4 096 sites 1.0x at 64 and 1.5x at
//    256, 1 728 sites 0.7x at 256 and 2.6x at 1 024, never at 1 024 sites; big graphs: 0.9x at 4 replicas, 1.5x at 8, 2.7x at 15;
//  * bit-sliced path (a thread decides FOUR positions: a quarter of the threads): resident-size graphs 0.8x at 4 096 sites x 256,
//    3.0x x 1 024; 0.67x at 8 000 x 256, 2.6x x 1 024; 13 824 sites 0.7x at 16 replicas, 1.2x at 64; 128^3: 1.4x at ONE replica.
static bool packed_worth_it(const isingmc_graph *g, size_t n_replicas, bool real_path)
{
    const uint64_t work = uint64_t(g->nvars) * n_replicas; // attempts per timestep
    const bool csr_resident = gen_resident_fits(g, n_replicas); // (the A/B switch ISINGMC_DISABLE_RESIDENT changes kernels, never the family)
    if (real_path) {
        // partial groups draw only their own replicas' Philox calls: 1.09x the CSR launches at ONE experiment (2048^2 Gaussian glass,
        // profiles/r03_few_replicas.txt; re-measured in profiles/r04_real_eligibility.txt), 1.4x at 2, 2.0x at 4, 2.9x at 8
        if (!csr_resident) return n_replicas >= 1;
        return n_replicas >= 16 && (g->nvars >= 8000 || (g->nvars >= 1500 && work >= (uint64_t(3) << 19))); // 1.5 x 2^20: re-measured with the graph staged in LDS
    }
    return work >= (uint64_t(1) << (csr_resident ? 22 : 19));
}

// 0: one replica per word set (CSR kernels); 1: replica-packed bit-sliced path (S6); 2: replica-packed real-coupling path (S7)
static int choose_packed(const isingmc_states *s, size_t n_replicas)
{
    const isingmc_graph *g = s->g;
    const Options &o = s->opt;
    if (g->rj_ok && !o.disable_real && (!g->packed_ok || o.force_real)) {
        // (stable_path: the family follows from the graph alone -- experiment k must not change when the call asks for more of them)
        if (o.force_real || g->stable_path) return n_replicas > 0 ? 2 : 0;
        return packed_worth_it(g, n_replicas, true) ? 2 : 0;
    }
    if (!g->packed_ok || o.disable_packed) return 0;
    // pk_sweep_kernel addresses the ELL table through one buffer descriptor with 32-bit byte offsets
    if (uint64_t(g->pk.n_pos) * PK_MAX_DEG * sizeof(uint32_t) >= (uint64_t(1) << 31)) return 0;
    if (o.force_packed || g->stable_path) return n_replicas > 0 ? 1 : 0;
    return packed_worth_it(g, n_replicas, false) ? 1 : 0;
}

// threshold table of one group for per-replica betas (beta_of(r) for r = 0..31)
template <typename F>
static void pk_fill_table(uint32_t *tab, double jabs, F &&beta_of)
{
    std::fill(tab, tab + PK_TAB_WORDS, 0u);
    for (uint32_t m = 1; m <= uint32_t(PK_MAX_DEG); m++)
        for (uint32_t r = 0; r < 32; r++) {
            const uint64_t T = threshold_fixed(beta_of(r), 2.0 * jabs * double(m));
            if (T >> THR_BITS) tab[PK_TAB_ALL + m - 1] |= 1u << r;
            const uint32_t hi = uint32_t(T >> 32) & ((1u << N_PLANES) - 1);
            for (int p = 0; p < N_PLANES; p++)
                if ((hi >> (N_PLANES - 1 - p)) & 1u) tab[PK_TAB_TBW + (m - 1) * N_PLANES + p] |= 1u << r;
            tab[PK_TAB_LO + (m - 1) * 32 + r] = uint32_t(T);
        }
    // (meaningful when every replica has the same beta: the one-degree kernel's uniform-beta instantiation reads them)
    for (uint32_t idx = 0; idx < uint32_t(PK_MAX_DEG) * N_PLANES; idx++)
        if (tab[PK_TAB_TBW + idx] & 1u) tab[PK_TAB_SEL + (idx >> 5)] |= 1u << (idx & 31);
}

static int pk_create(isingmc_states *s, const uint64_t *all_seeds, size_t first, size_t n, const uint8_t *initial_state)
{
    const isingmc_graph *g = s->g;
    s->packed = true;
    const size_t group0 = first / 32; // global groups [group0, group0 + groups) intersect this shard
    s->pk_bit0 = first % 32;
    s->groups = (s->pk_bit0 + n + 31) / 32;
    s->R = s->cap = n;
    TRY(dev_alloc(&s->d_state, s->groups * g->pk.n_pos));
    TRY(dev_alloc(&s->d_keys, s->groups));
    TRY(dev_alloc(&s->d_meas, 2 * s->pk_slots()));
    s->meas_zero = false;
    std::vector<uint2> keys(s->groups);
    for (size_t k = 0; k < s->groups; k++) { // a group is keyed by the seed of its first GLOBAL replica
        const uint64_t seed = all_seeds[32 * (group0 + k)];
        keys[k] = make_uint2(uint32_t(seed), uint32_t(seed >> 32));
    }
    HIP_TRY(hipMemcpy(s->d_keys, keys.data(), keys.size() * sizeof(uint2), hipMemcpyHostToDevice));
    if (initial_state) { // every replica starts from the same configuration: a word is all ones or all zeros
        std::vector<uint32_t> words(g->pk.n_pos, 0u);
        for (uint64_t i = 0; i < g->nvars; i++) words[g->pos[i]] = initial_state[i] ? 0xFFFFFFFFu : 0u;
        for (size_t k = 0; k < s->groups; k++)
            HIP_TRY(hipMemcpy(s->d_state + k * g->pk.n_pos, words.data(), words.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    } else {
        for (size_t g0 = 0; g0 < s->groups; g0 += MAX_GRID_Y) {
            const size_t ng = std::min(MAX_GRID_Y, s->groups - g0);
            hipLaunchKernelGGL(pk_init_kernel, dim3((g->pk.n_pos + 255) / 256, unsigned(ng)), dim3(256), 0, s->stream, s->d_state,
                               g->pk, s->d_keys, uint32_t(g0));
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    return ISINGMC_OK;
}

// ClassicIsing.add_graph (classicising.rs:62-79) on a replica-packed container.  A replica appended into a partly filled
// group starts from the random start of its bit position (the "PKIN" words of the group's key) NOW, at the current timestep --
// a fresh experiment, as GraphState::new gives the reference (classicising.rs:73) and as every other path here does; with an
// initial_state it is set explicitly.  (Bit-sliced path: a group simulates all of its 32 bit positions from the moment it is
// created -- the spec numbers ties over whole groups -- so until round 3 the new replica took over the chain its bit had been
// running: a thermalised start where the caller asked for a random one.)  Replica 32 g opens a new group keyed by its seed,
// randomly started now.  Only whole containers grow (not shards of a larger set of experiments).
static int pk_append(isingmc_states *s, uint64_t seed, const uint8_t *initial_state)
{
    const isingmc_graph *g = s->g;
    if (s->first != 0 || s->n_total != s->R) return fail(ISINGMC_ERR_INVALID, "a shard of a larger set of experiments cannot grow");
    const size_t slot = s->R;
    if (slot % 32 == 0) { // a new group
        const size_t groups = s->groups + 1;
        uint32_t *d_state = nullptr;
        uint2 *d_keys = nullptr;
        unsigned long long *d_meas = nullptr;
        TRY(dev_alloc(&d_state, groups * g->pk.n_pos));
        struct Undo { void *a, **b, **c; bool armed = true; ~Undo() { if (armed) { (void)cached_free(a); if (*b) (void)cached_free(*b); if (*c) (void)cached_free(*c); } } }
            undo{d_state, reinterpret_cast<void **>(&d_keys), reinterpret_cast<void **>(&d_meas)};
        TRY(dev_alloc(&d_keys, groups));
        TRY(dev_alloc(&d_meas, 2 * 32 * groups));
        HIP_TRY(hipStreamSynchronize(s->stream));
        HIP_TRY(hipMemcpy(d_state, s->d_state, s->groups * g->pk.n_pos * sizeof(uint32_t), hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(d_keys, s->d_keys, s->groups * sizeof(uint2), hipMemcpyDeviceToDevice));
        HIP_TRY(hipDeviceSynchronize()); // as in reserve(): the old blocks go back to the cache below
        const uint2 key = make_uint2(uint32_t(seed), uint32_t(seed >> 32));
        HIP_TRY(hipMemcpy(d_keys + s->groups, &key, sizeof key, hipMemcpyHostToDevice));
        undo.armed = false;
        (void)cached_free(s->d_state); (void)cached_free(s->d_keys); (void)cached_free(s->d_meas);
        s->d_state = d_state; s->d_keys = d_keys; s->d_meas = d_meas;
        s->meas_zero = false;
        if (s->d_tab) { (void)cached_free(s->d_tab); s->d_tab = nullptr; }             // per-group tables: rebuilt by the next set_betas
        if (s->d_pk_philox) { (void)cached_free(s->d_pk_philox); s->d_pk_philox = nullptr; s->pk_philox_steps = 0; } // rows per group count
        if (s->d_rj_betas) { (void)cached_free(s->d_rj_betas); s->d_rj_betas = nullptr; }
        hipLaunchKernelGGL(pk_init_kernel, dim3((g->pk.n_pos + 255) / 256, 1), dim3(256), 0, s->stream, s->d_state, g->pk, s->d_keys,
                           uint32_t(s->groups));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s->stream));
        s->groups = groups;
    }
    s->R = s->cap = s->n_total = slot + 1;
    if (initial_state) TRY(pk_set_state(s, slot, initial_state));
    else if (slot % 32 != 0) {
        // (real-coupling path: the bits of a group this container does not own are not simulated at all, rj_sweep_kernel PARTIAL,
        //  so the column holds whatever it held)
        hipLaunchKernelGGL(pk_init_replica_kernel, dim3((g->pk.n_pos + 255) / 256), dim3(256), 0, s->stream, s->d_state, g->pk, s->d_keys,
                           uint32_t(slot / 32), uint32_t(slot % 32));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    return ISINGMC_OK;
}

static int pk_set_state(isingmc_states *s, size_t replica, const uint8_t *spins)
{
    const isingmc_graph *g = s->g;
    std::vector<uint32_t> bits(g->state_words, 0u);
    for (uint64_t i = 0; i < g->nvars; i++)
        if (spins[i]) bits[g->pos[i] >> 5] |= 1u << (g->pos[i] & 31);
    DeviceScratch scratch(s->stream);
    uint32_t *d_bits = nullptr;
    TRY(scratch.alloc(&d_bits, bits.size()));
    HIP_TRY(hipMemcpy(d_bits, bits.data(), bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pk_set_replica_kernel, dim3((g->pk.n_pos + 255) / 256), dim3(256), 0, s->stream, s->d_state,
                       g->pk.n_pos, d_bits, uint32_t((replica + s->pk_bit0) / 32), uint32_t((replica + s->pk_bit0) % 32));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    return ISINGMC_OK;
}

static int pk_set_betas(isingmc_states *s)
{
    if (s->rj) { // one acceptance scale per (group, bit); bits this shard does not own take the nearest owned replica's
        std::vector<RjBeta> tab(32 * s->groups);
        for (size_t sl = 0; sl < tab.size(); sl++) {
            const size_t r = sl < s->pk_bit0 ? 0 : std::min(s->R - 1, sl - s->pk_bit0);
            rj_beta(s->betas[r], s->g->rj_k, &tab[sl].shift, &tab[sl].mant);
        }
        if (!s->d_rj_betas) TRY(dev_alloc(&s->d_rj_betas, tab.size()));
        HIP_TRY(hipStreamSynchronize(s->stream)); // no launch may still be reading the old scales
        HIP_TRY(hipMemcpy(s->d_rj_betas, tab.data(), tab.size() * sizeof(RjBeta), hipMemcpyHostToDevice));
        return ISINGMC_OK;
    }
    std::vector<uint32_t> tabs(s->groups * PK_TAB_WORDS);
    for (size_t k = 0; k < s->groups; k++)
        pk_fill_table(tabs.data() + k * PK_TAB_WORDS, s->g->jabs,
                      [&](uint32_t r) { return s->betas[std::min(s->R - 1, 32 * k + r)]; }); // pk_bit0 == 0 (checked by the caller)
    if (!s->d_tab) TRY(dev_alloc(&s->d_tab, tabs.size())); // groups is fixed for the life of a packed container (no append): allocated once
    HIP_TRY(hipStreamSynchronize(s->stream)); // no launch may still be reading the old tables
    HIP_TRY(hipMemcpy(s->d_tab, tabs.data(), tabs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    return ISINGMC_OK;
}

// real-coupling path: one launch per colour class; a workgroup walks several 256-position blocks (it loads the log table once)
static void rj_launch_timestep(isingmc_states *s, const RjBeta *betas, uint32_t beta_stride, size_t gb, size_t ge, hipStream_t stream)
{
    const isingmc_graph *g = s->g;
    const int target_wgs = s->opt.real_target_wgs;
    for (uint32_t c = 0; c < g->n_colours; c++) {
        const uint32_t b = uint32_t(g->class_base[c]), e = g->class_real_end[c];
        if (e == b) continue;
        const size_t threads = rj_threads(g->rj.slots);
        const size_t nblocks = (size_t(e - b) + threads - 1) / threads;
        // The first and the last group of the container may own only part of their 32 replica bits (few experiments, a shard
        // cut inside a group): those groups get launches of their own that draw only the Philox calls of the owned bits.
        const auto quads_of = [&](size_t grp, uint32_t *lo, uint32_t *hi) {
            const size_t lo_bit = grp == 0 ? s->pk_bit0 : 0;
            const size_t hi_bit = grp + 1 == s->groups ? (s->pk_bit0 + s->R - 1) % 32 + 1 : 32;
            *lo = uint32_t(lo_bit / 4);
            *hi = uint32_t((hi_bit + 3) / 4);
        };
        for (size_t g0 = gb; g0 < ge;) {
            uint32_t q_lo, q_hi;
            quads_of(g0, &q_lo, &q_hi);
            size_t ng = 1; // extend over the following groups with the same quads (all the whole groups in the middle)
            for (uint32_t l2, h2; g0 + ng < ge && ng < MAX_GRID_Y; ng++) {
                quads_of(g0 + ng, &l2, &h2);
                if (l2 != q_lo || h2 != q_hi) break;
            }
            const size_t gx0 = std::min(nblocks, std::max<size_t>(1, (size_t(target_wgs) + ng - 1) / ng));
            const size_t per = (nblocks + gx0 - 1) / gx0, gx = (nblocks + per - 1) / per; // equal shares, no short last round
            (void)rj_launch_sweep(dim3(unsigned(gx), unsigned(ng)), stream, s->d_state + g0 * g->pk.n_pos, g->rj, b, e, s->t,
                                  s->d_keys + g0, betas + (beta_stride ? g0 * beta_stride : 0), beta_stride, q_lo, q_hi);
            g0 += ng;
        }
    }
}

static void pk_launch_timestep(isingmc_states *s, const uint32_t *tabs, uint32_t tab_stride, const uint32_t *philox_tab, size_t gb, size_t ge,
                               hipStream_t stream)
{
    const isingmc_graph *g = s->g;
    const bool no_uni = s->opt.disable_packed_uniform != 0; // A/B switch: results are the same either way
    for (uint32_t c = 0; c < g->n_colours; c++) {
        const uint32_t b = uint32_t(g->class_base[c]), e = uint32_t(g->class_base[c + 1]);
        if (e == b) continue;
        // blocks of real sites of a one-degree graph: the specialised kernel; the class's padded tail (and
        // every other graph): the general one.  tab_stride == 0 <=> one table, one beta for every replica.
        const uint32_t mid = g->pk_uni_deg && !no_uni ? g->pk_class_full[c] : b;
        for (size_t g0 = gb; g0 < ge; g0 += MAX_GRID_Y) {
            const size_t ng = std::min(MAX_GRID_Y, ge - g0);
            if (mid > b)
                (void)pk_uni_launch_sweep(g->pk_uni_deg, tab_stride == 0, g->pk_uni_pmj, uint32_t(ng), stream,
                                          s->d_state + g0 * g->pk.n_pos, g->pk, g->pk_uni, b, mid, s->t, s->d_keys + g0,
                                          tabs + (tab_stride ? g0 * tab_stride : 0), tab_stride,
                                          philox_tab + g0 * pk_uni_philox_table_words(), g->pk_class_table[c] != 0);
            if (e > mid)
                hipLaunchKernelGGL(pk_sweep_kernel, dim3((e - mid) / 1024 + ((e - mid) % 1024 != 0), unsigned(ng)), dim3(256), 0, stream,
                                   s->d_state + g0 * g->pk.n_pos, g->pk, mid, e, s->t, s->d_keys + g0,
                                   tabs + (tab_stride ? g0 * tab_stride : 0), tab_stride);
        }
    }
}


// energy of one replica of a packed container from the counters of its slot
double pk_energy(const isingmc_graph *g, bool rj, unsigned long long c0, unsigned long long c1)
{
    // real-coupling path: c0, c1 = -2 x the hi / lo level sums of the energy (exact, even integers; rj_measure_kernel, run once
    // per level): E = (2^kE hi + 2^(kE - 24) lo) + self loops -- the energy of the ORIGINAL couplings to Fmax 2^-54 per term
    if (rj) return (std::ldexp(double(-(int64_t(c0) / 2)), g->rj_k_energy) + std::ldexp(double(-(int64_t(c1) / 2)), g->rj_k_energy - RJ_ENERGY_LO_BITS)) + g->self_energy;
    (void)c1;
    // bit-sliced path: E = |J| (undirected bonds - 2 satisfied) + self loops; c0 = directed satisfied count (doubled)
    return g->jabs * (double(int64_t(g->n_directed / 2)) - double(int64_t(c0))) + g->self_energy;
}

static int pk_measure(isingmc_states *s, double *energies, int64_t *mags)
{
    const isingmc_graph *g = s->g;
    const size_t R = s->R;
    if (s->rj && energies && mags) { // the second counter of a slot holds the lo level of the energy OR the up spins
        TRY(pk_measure(s, energies, nullptr));
        return pk_measure(s, nullptr, mags);
    }
    TRY(measure_enqueue(s, s->d_meas, nullptr, nullptr, /*want_up=*/mags != nullptr));
    s->meas_zero = false;
    std::vector<unsigned long long> h(2 * s->pk_slots());
    HIP_TRY(hipMemcpyAsync(h.data(), s->d_meas, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (size_t r = 0; r < R; r++) {
        const size_t sl = r + s->pk_bit0;
        if (energies) energies[r] = pk_energy(g, s->rj, h[2 * sl], h[2 * sl + 1]);
        if (mags) mags[r] = 2 * int64_t(h[2 * sl + 1]) - int64_t(g->nvars);
    }
    return ISINGMC_OK;
}

static int pk_run_steps(isingmc_states *s, size_t timesteps, const double *betas, size_t beta_stride,
                        double *energies_per_step, float *device_ms, bool sync)
{
    const isingmc_graph *g = s->g;
    const size_t R = s->R, CS = s->pk_slots();
    DeviceScratch scratch(s->stream);
    uint32_t *d_step_tabs = nullptr;
    RjBeta *d_rj_steps = nullptr; // real-coupling path: one acceptance scale per timestep of the chunk
    std::vector<RjBeta> h_rj;
    const size_t chunk = std::min<size_t>(timesteps, 2048);
    if (!s->has_betas && s->rj) TRY(scratch.alloc(&d_rj_steps, beta_stride ? chunk : 1));
    else if (!s->has_betas) TRY(scratch.alloc(&d_step_tabs, (beta_stride ? chunk : 1) * PK_TAB_WORDS));
    // energies after every timestep: the measurements are enqueued behind their sweeps into one counter slot
    // per step; the host reads a whole chunk at once
    unsigned long long *d_step_counts = nullptr;
    std::vector<unsigned long long> h_step_counts;
    if (energies_per_step) {
        TRY(scratch.alloc(&d_step_counts, chunk * CS * 2));
        h_step_counts.resize(chunk * CS * 2);
    }
    std::vector<uint32_t> h_tabs;
    // one-degree kernels: the wave-uniform halves of every timestep's Philox calls, PK_PHILOX_STEPS timesteps ahead, written by one
    // small launch when the rows run out (s->d_pk_philox: kept across calls)
    constexpr size_t PK_PHILOX_STEPS = 2048; // (>= chunk)
    const size_t philox_words = !s->rj && g->pk_uni_deg && !s->opt.disable_packed_uniform ? s->groups * pk_uni_philox_table_words() : 0;
    const auto philox_rows_ready = [&](size_t nk) {
        return s->pk_philox_steps && s->pk_philox_groups == s->groups && s->t >= s->pk_philox_t0 &&
               s->t + nk <= s->pk_philox_t0 + s->pk_philox_steps;
    };
    int rc = ISINGMC_OK;
    if (device_ms) HIP_TRY(hipEventRecord(s->ev0, s->stream));
    // The replica groups are independent: mid-size launches (a few waves per SIMD: the 64^3 glass x 64 replicas puts ONE wave on a
    // SIMD per colour-class launch) leave the chip idle around every kernel boundary, so the groups go to several streams and one
    // lane's launch gap / ramp / tail overlaps the other lanes' work (as the lattice path's replica lanes).  ISINGMC_PK_STREAMS=<n> forces.
    size_t want_lanes = 1;
    if (!energies_per_step && s->groups >= 2) {
        uint64_t biggest = 0;
        for (uint32_t c = 0; c < g->n_colours; c++) biggest = std::max<uint64_t>(biggest, g->class_base[c + 1] - g->class_base[c]);
        const uint64_t waves_per_launch = s->groups * biggest / (s->rj ? 64 : 256); // a thread decides 1 (real) / 4 (bit-sliced) positions
        const int forced = s->opt.pk_streams;
        if (forced > 0) want_lanes = size_t(forced);
        // measured (tools/pk_lanes_ab.py, profiles/r03_pk_lanes_ab.txt): two lanes +7 % (2048^2 x 256) to +43 % (512^2 x 64) from ~2 000 waves per
        // launch on, -3..-13 % below (32^3 x 64: the launches are too short for the fork / join); four lanes: worse than two almost everywhere
        // Short calls (the 10-timestep blocks between tempering rounds) double their launch count with lanes and run into the host's
        // launch rate sooner: 64^3 x 64 rungs went from 24.5 to 31 us per timestep; they take lanes only for long launches
        else if (timesteps >= 64 ? waves_per_launch >= 2048 : timesteps >= 4 && waves_per_launch >= 16384) want_lanes = 2;
        want_lanes = std::min(want_lanes, s->groups);
    }
    struct LaneJoin {
        isingmc_states *s;
        ~LaneJoin() { if (s->n_lanes > 1) (void)lanes_join(s); }
    } lane_join{s};
    const size_t n_lanes = want_lanes, per_lane = (s->groups + n_lanes - 1) / n_lanes;
    const auto launch_step = [&](size_t k) {
        for (size_t lane = 0; lane < n_lanes; lane++) {
            const size_t gb = lane * per_lane, ge = std::min(s->groups, gb + per_lane);
            if (gb >= ge) continue;
            hipStream_t st = n_lanes > 1 ? s->lanes[lane] : s->stream;
            if (s->rj) {
                if (s->has_betas) rj_launch_timestep(s, s->d_rj_betas, 32, gb, ge, st);
                else rj_launch_timestep(s, d_rj_steps + (beta_stride ? k : 0), 0, gb, ge, st);
            } else {
                const uint32_t *rows = philox_words ? s->d_pk_philox + size_t(s->t - s->pk_philox_t0) * philox_words : nullptr; // timestep s->t's
                if (s->has_betas) pk_launch_timestep(s, s->d_tab, PK_TAB_WORDS, rows, gb, ge, st);
                else pk_launch_timestep(s, d_step_tabs + (beta_stride ? k * PK_TAB_WORDS : 0), 0, rows, gb, ge, st);
            }
        }
    };
    for (size_t k0 = 0; k0 < timesteps && rc == ISINGMC_OK; k0 += chunk) {
        const size_t nk = std::min(chunk, timesteps - k0);
        const bool new_step_tabs = !s->has_betas && (beta_stride || k0 == 0);
        const bool new_philox_rows = philox_words && !philox_rows_ready(nk);
        if (k0 > 0 && (new_step_tabs || new_philox_rows)) { // the chunk's tables are overwritten: every lane must have finished reading them
            if (s->n_lanes > 1) TRY(lanes_join(s));
            if (new_step_tabs) HIP_TRY(hipStreamSynchronize(s->stream)); // (written from the host)
        }
        if (!s->has_betas && s->rj && (beta_stride || k0 == 0)) {
            h_rj.resize(beta_stride ? nk : 1);
            for (size_t k = 0; k < h_rj.size(); k++) rj_beta(betas[(k0 + k) * beta_stride], g->rj_k, &h_rj[k].shift, &h_rj[k].mant);
            HIP_TRY(hipMemcpy(d_rj_steps, h_rj.data(), h_rj.size() * sizeof(RjBeta), hipMemcpyHostToDevice));
        } else if (!s->has_betas && (beta_stride || k0 == 0)) {
            h_tabs.resize((beta_stride ? nk : 1) * PK_TAB_WORDS);
            for (size_t k = 0; k < h_tabs.size() / PK_TAB_WORDS; k++) {
                const double beta = betas[(k0 + k) * beta_stride];
                pk_fill_table(h_tabs.data() + k * PK_TAB_WORDS, g->jabs, [&](uint32_t) { return beta; });
            }
            HIP_TRY(hipMemcpy(d_step_tabs, h_tabs.data(), h_tabs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        if (new_philox_rows) { // (on the main stream, the lanes joined: in order with every sweep that read the old rows)
            if (!s->d_pk_philox) TRY(dev_alloc(&s->d_pk_philox, PK_PHILOX_STEPS * philox_words));
            HIP_TRY(pk_uni_launch_philox_table(s->stream, s->d_pk_philox, s->d_keys, uint32_t(s->groups), s->t, uint32_t(PK_PHILOX_STEPS)));
            s->pk_philox_t0 = s->t;
            s->pk_philox_steps = PK_PHILOX_STEPS;
            s->pk_philox_groups = s->groups;
        }
        if (want_lanes > 1 && s->n_lanes <= 1) TRY(lanes_fork(s, want_lanes)); // (behind the table launch: the lanes wait for it)
        for (size_t k = 0; k < nk && rc == ISINGMC_OK; k++) {
            launch_step(k);
            s->t++;
            if (energies_per_step) rc = measure_enqueue(s, d_step_counts + k * CS * 2, nullptr, nullptr, /*want_up=*/false);
        }
        if (energies_per_step && rc == ISINGMC_OK) {
            HIP_TRY(hipMemcpyAsync(h_step_counts.data(), d_step_counts, nk * CS * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
            HIP_TRY(hipStreamSynchronize(s->stream));
            for (size_t k = 0; k < nk; k++)
                for (size_t r = 0; r < R; r++)
                    energies_per_step[r * timesteps + k0 + k] = pk_energy(g, s->rj, h_step_counts[(k * CS + r + s->pk_bit0) * 2],
                                                                          h_step_counts[(k * CS + r + s->pk_bit0) * 2 + 1]);
        }
    }
    if (s->n_lanes > 1) { const int jrc = lanes_join(s); if (rc == ISINGMC_OK) rc = jrc; }
    if (device_ms && rc == ISINGMC_OK) {
        hipError_t err = hipEventRecord(s->ev1, s->stream);
        if (err == hipSuccess) err = hipEventSynchronize(s->ev1);
        if (err == hipSuccess) err = hipEventElapsedTime(device_ms, s->ev0, s->ev1);
        if (err != hipSuccess) rc = fail(ISINGMC_ERR_HIP, hipGetErrorString(err));
    }
    if (rc == ISINGMC_OK) {
        hipError_t err = hipGetLastError();
        if (err == hipSuccess && sync) err = hipStreamSynchronize(s->stream);
        if (err != hipSuccess) rc = fail(ISINGMC_ERR_HIP, hipGetErrorString(err));
    }
    return rc; // `scratch` drains the stream before it frees the step tables
}

// packed words -> one byte per spin, replica by replica
int pk_get_states(isingmc_states *s, uint8_t *states_out, size_t replica_stride_bytes, uint32_t *packed_out)
{
    const isingmc_graph *g = s->g;
    std::vector<uint32_t> words(s->groups * g->pk.n_pos);
    HIP_TRY(hipMemcpyAsync(words.data(), s->d_state, words.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    parallel_for(s->R, [&](size_t r) {
        const uint32_t *w = words.data() + ((r + s->pk_bit0) / 32) * g->pk.n_pos;
        const uint32_t bit = uint32_t((r + s->pk_bit0) % 32);
        if (states_out) {
            uint8_t *out = states_out + r * replica_stride_bytes;
            for (uint64_t i = 0; i < g->nvars; i++) out[i] = (w[g->pos[i]] >> bit) & 1u;
        }
        if (packed_out) { // the per-replica layout of the thread-per-site path (bit-packed by position)
            uint32_t *out = packed_out + r * g->state_words;
            std::fill(out, out + g->state_words, 0u);
            for (uint64_t i = 0; i < g->nvars; i++)
                out[g->pos[i] >> 5] |= ((w[g->pos[i]] >> bit) & 1u) << (g->pos[i] & 31);
        }
    });
    return ISINGMC_OK;
}

// ------------------------------------------------------------------------------------------------
// sweeps
// ------------------------------------------------------------------------------------------------
template <bool VEC, bool PMJ>
static void launch_lat_sweep(isingmc_states *s, uint32_t colour, const LatThr &thr, uint64_t t_arg)
{
    const isingmc_graph *g = s->g;
    // replicas are independent: with n_lanes > 1 the replica blocks go to different streams, so that the
    // launch gap / ramp / tail of one block's half-sweep overlaps the other blocks' work
    const size_t per_lane = (s->R + s->n_lanes - 1) / s->n_lanes;
    for (size_t lane = 0; lane < s->n_lanes; lane++) {
        const size_t lo = lane * per_lane, hi = std::min(s->R, lo + per_lane);
        hipStream_t stream = s->n_lanes > 1 ? s->lanes[lane] : s->stream;
        for (size_t r0 = lo; r0 < hi; r0 += MAX_GRID_Y) {
            const size_t n = std::min(MAX_GRID_Y, hi - r0);
            // diagnostic: ISINGMC_DEBUG_SWEEP_LDS=<bytes> of unused LDS per workgroup lowers the occupancy
            const unsigned dbg_lds = unsigned(std::max(0, s->opt.debug_sweep_lds));
            const auto launch = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, lat_grid(g, g->geom.nquads, n), dim3(256), dbg_lds, stream,
                                   s->d_state + r0 * g->state_words, g->geom, colour, t_arg, s->d_keys + r0, thr,
                                   s->has_betas ? s->d_thr + r0 : nullptr, g->d_jneg, g->jneg_uniform);
            };
            // large launches: every thread loops over two quads (measured: 2 quads +4 %, 4 +2.7 %, 8 +1.3 % on
            // uniform J; +-J: 2 quads +2.6 %, 4 quads -2 %).  Workgroups of 128 or 64 threads: no gain.
            // ISINGMC_SWEEP_ITERS=1|2|4|8 forces the choice (measurement only)
            uint32_t iters = 1;
            if (VEC && g->geom.cols_log2 >= 0) {
                const int forced = s->opt.sweep_iters;
                const uint32_t want = forced ? uint32_t(forced) : 2u;
                if (want > 1 && g->geom.nquads % (256 * want) == 0 &&
                    (forced || size_t(g->geom.nquads / (256 * want)) * n >= size_t(8) * 256)) // >= 8 workgroups per CU left (c4: +2.6 %)
                    iters = want;
            }
            if (iters > 1)
                hipLaunchKernelGGL(lat_sweep_loop_kernel<PMJ>, dim3(g->geom.nquads / (256 * iters), unsigned(n), 1), dim3(256), dbg_lds, stream,
                                   s->d_state + r0 * g->state_words, g->geom, colour, t_arg, s->d_keys + r0, thr,
                                   s->has_betas ? s->d_thr + r0 : nullptr, g->d_jneg, g->jneg_uniform, iters);
            else if (VEC && g->geom.cols_log2 >= 0) launch(lat_sweep_kernel<VEC, PMJ, VEC>); // division-free mapping
            else launch(lat_sweep_kernel<VEC, PMJ, false>);
        }
    }
}

// fork: the lanes wait for everything queued on the main stream; join: the main stream waits for the lanes
static int lanes_reserve(isingmc_states *s, size_t n)
{
    while (s->lanes.size() < n) {
        hipStream_t st;
        hipEvent_t ev;
        HIP_TRY(pooled_stream_create(&st));
        HIP_TRY(pooled_event_create(&ev, true));
        s->lanes.push_back(st);
        s->lane_events.push_back(ev);
    }
    if (!s->fork_event) HIP_TRY(pooled_event_create(&s->fork_event, true));
    return ISINGMC_OK;
}

static int lanes_fork(isingmc_states *s, size_t n)
{
    TRY(lanes_reserve(s, n));
    HIP_TRY(hipEventRecord(s->fork_event, s->stream));
    for (size_t i = 0; i < n; i++) HIP_TRY(hipStreamWaitEvent(s->lanes[i], s->fork_event, 0));
    s->n_lanes = n;
    return ISINGMC_OK;
}

static int lanes_join(isingmc_states *s)
{
    for (size_t i = 0; i < s->n_lanes && s->n_lanes > 1; i++) {
        HIP_TRY(hipEventRecord(s->lane_events[i], s->lanes[i]));
        HIP_TRY(hipStreamWaitEvent(s->stream, s->lane_events[i], 0));
    }
    s->n_lanes = 1;
    return ISINGMC_OK;
}

template <bool VEC, bool PMJ>
static void launch_lat_measure(isingmc_states *s, unsigned long long *out, size_t out_stride)
{
    const isingmc_graph *g = s->g;
    for (size_t r0 = 0; r0 < s->R; r0 += MAX_GRID_Y) {
        const size_t n = std::min(MAX_GRID_Y, s->R - r0);
        const uint32_t blocks = (g->geom.nquads + 256 * MEASURE_QUADS_PER_THREAD - 1) / (256 * MEASURE_QUADS_PER_THREAD);
        if (g->mc_mode == MC_ANISO) { // the two directions' bonds carry different |J|: counted apart
            (void)mc_launch_measure_aniso(PMJ, dim3(blocks, unsigned(n)), s->stream, s->d_state + r0 * g->state_words, g->geom, g->d_jneg,
                                          g->jneg_uniform, out + r0 * out_stride, out_stride);
            continue;
        }
        if (g->mc_mode == MC_OPEN || g->mc_mode == MC_FIELD_OPEN || g->d_fneg) {
            // the bonds across an open boundary do not exist: they must not count as satisfied; with field-sign planes the
            // spins along their site's field are counted too
            (void)mc_launch_measure_open(PMJ, dim3(blocks, unsigned(n)), s->stream, s->d_state + r0 * g->state_words, g->geom, g->d_jneg,
                                         g->jneg_uniform, g->open, g->d_fneg, out + r0 * out_stride, out_stride);
            continue;
        }
        hipLaunchKernelGGL((lat_measure_kernel<VEC, PMJ>), dim3(blocks, unsigned(n)), dim3(256), 0, s->stream,
                           s->d_state + r0 * g->state_words, g->geom, g->d_jneg, g->jneg_uniform,
                           out + r0 * out_stride, out_stride);
    }
}

// colour-1 half-sweep fused with the measurement of the finished timestep (single stream: the per-step energy
// mode does not use replica lanes)
template <bool VEC, bool PMJ>
static void launch_lat_sweep_measure(isingmc_states *s, const LatThr &thr, uint64_t t_arg, unsigned long long *out, size_t out_stride)
{
    const isingmc_graph *g = s->g;
    for (size_t r0 = 0; r0 < s->R; r0 += MAX_GRID_Y) {
        const size_t n = std::min(MAX_GRID_Y, s->R - r0);
        const auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, lat_grid(g, g->geom.nquads, n), dim3(256), 0, s->stream, s->d_state + r0 * g->state_words,
                               g->geom, t_arg, s->d_keys + r0, thr, s->has_betas ? s->d_thr + r0 : nullptr, g->d_jneg,
                               g->jneg_uniform, out + r0 * out_stride, out_stride);
        };
        if (VEC && g->geom.cols_log2 >= 0) launch(lat_sweep_measure_kernel<VEC, PMJ, VEC>);
        else launch(lat_sweep_measure_kernel<VEC, PMJ, false>);
    }
}

#define LAT_DISPATCH(fn, ...)                                                                       \
    do {                                                                                            \
        const bool pmj__ = !s->g->uniform_sign;                                                     \
        if (s->g->vec) { if (pmj__) fn<true, true>(__VA_ARGS__); else fn<true, false>(__VA_ARGS__); } \
        else { if (pmj__) fn<false, true>(__VA_ARGS__); else fn<false, false>(__VA_ARGS__); }       \
    } while (0)

#ifndef ISINGMC_GEN_RB
#define ISINGMC_GEN_RB 8
#endif
constexpr int GEN_RB = ISINGMC_GEN_RB; // replicas per thread on the general path (amortises the CSR stream)

template <typename WT, int RB>
static void launch_gen_class(isingmc_states *s, uint32_t b, uint32_t e, double beta)
{
    const isingmc_graph *g = s->g;
    const size_t chunk = MAX_GRID_Y * RB;
    for (size_t r0 = 0; r0 < s->R; r0 += chunk) {
        const size_t n = std::min(chunk, s->R - r0);
        const dim3 grid((e - b + 255) / 256, unsigned((n + RB - 1) / RB));
        hipLaunchKernelGGL((gen_sweep_kernel<WT, RB>), grid, dim3(256), 0, s->stream, s->d_state + r0 * g->state_words,
                           g->gdev, b, e, s->t, s->d_keys + r0, beta, s->has_betas ? s->d_beta + r0 : nullptr, uint32_t(n));
    }
}

static void launch_gen_timestep(isingmc_states *s, double beta)
{
    const isingmc_graph *g = s->g;
    for (uint32_t c = 0; c < g->n_colours; c++) {
        const uint32_t b = uint32_t(g->class_base[c]), e = uint32_t(g->class_base[c + 1]);
        if (e == b) continue;
        // replicas per thread: GEN_RB amortises the CSR stream of a big class; a class that would leave the chip
        // short of workgroups (< 8 per CU) halves it until the grid is large enough
        // (200 000 sites x 64 replicas: 56 us per launch at 8 replicas per thread)
        size_t rb = GEN_RB;
        const size_t blocks = (e - b + 255) / 256;
        while (rb > 1 && (s->R < rb || blocks * ((s->R + rb - 1) / rb) < 2048)) rb /= 2;
        const auto launch = [&](auto wt) {
            using WT = decltype(wt);
            if (rb >= 8) launch_gen_class<WT, 8>(s, b, e, beta);
            else if (rb == 4) launch_gen_class<WT, 4>(s, b, e, beta);
            else if (rb == 2) launch_gen_class<WT, 2>(s, b, e, beta);
            else launch_gen_class<WT, 1>(s, b, e, beta);
        };
        if (g->w_is_float) launch(float(0)); else launch(double(0));
    }
}

// energies / magnetisations of the current configurations into host arrays (either may be NULL)
static int measure(isingmc_states *s, double *energies, int64_t *mags)
{
    const isingmc_graph *g = s->g;
    const size_t R = s->R;
    if (R == 0) return ISINGMC_OK;
    if (s->packed) return pk_measure(s, energies, mags);
    if (g->kind == ISINGMC_KIND_LATTICE2D) {
        HIP_TRY(hipMemsetAsync(s->d_meas, 0, 2 * R * sizeof(unsigned long long), s->stream));
        s->meas_zero = false;
        LAT_DISPATCH(launch_lat_measure, s, s->d_meas, size_t(2));
        HIP_TRY(hipGetLastError());
        std::vector<unsigned long long> h(2 * R);
        HIP_TRY(hipMemcpyAsync(h.data(), s->d_meas, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        for (size_t r = 0; r < R; r++) {
            if (energies) energies[r] = lattice_energy(g, h[2 * r], h[2 * r + 1]);
            if (mags) mags[r] = 2 * int64_t(h[2 * r + 1]) - int64_t(g->nvars);
        }
    } else {
        for (size_t r0 = 0; r0 < R; r0 += MAX_GRID_Y) {
            const size_t n = std::min(MAX_GRID_Y, R - r0);
            const dim3 grid(s->n_partials, unsigned(n));
            if (g->w_is_float)
                hipLaunchKernelGGL(gen_measure_kernel<float>, grid, dim3(256), 0, s->stream,
                                   s->d_state + r0 * g->state_words, g->gdev, s->d_pe + r0 * s->n_partials,
                                   s->d_pm + r0 * s->n_partials);
            else
                hipLaunchKernelGGL(gen_measure_kernel<double>, grid, dim3(256), 0, s->stream,
                                   s->d_state + r0 * g->state_words, g->gdev, s->d_pe + r0 * s->n_partials,
                                   s->d_pm + r0 * s->n_partials);
        }
        hipLaunchKernelGGL(gen_reduce_kernel, dim3(unsigned(R)), dim3(256), 0, s->stream, s->d_pe, s->d_pm,
                           s->n_partials, s->d_oe, s->d_om);
        HIP_TRY(hipGetLastError());
        std::vector<double> he(R);
        std::vector<long long> hm(R);
        HIP_TRY(hipMemcpyAsync(he.data(), s->d_oe, R * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipMemcpyAsync(hm.data(), s->d_om, R * sizeof(long long), hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        for (size_t r = 0; r < R; r++) {
            if (energies) energies[r] = he[r] + g->self_energy;
            if (mags) mags[r] = hm[r];
        }
    }
    return ISINGMC_OK;
}

// ISINGMC_DISABLE_RESIDENT=1 (at creation): always use the per-colour launches (A/B runs, parity tests of both paths)
static bool resident_disabled(const isingmc_states *s) { return s->opt.disable_resident != 0; }

// ------------------------------------------------------------------------------------------------
// persistent strip kernel (strip_kernels.hpp): when and how
// ------------------------------------------------------------------------------------------------

// Mid-size lattices only: a per-colour launch of the streaming kernel must be short enough for the ~5 us it loses
// between dependent launches to matter (<= ISINGMC_STRIP_MAX_WG workgroups in all, default the resident limit), the geometry must cut into strips of 256 quads with at least two strips per replica, and the poll of a
// half-sweep must fit one workgroup (2 rows of <= 128 words).  ISINGMC_STRIP=0 disables, =1 forces (tests, A/B runs).
// resident workgroups per CU, cached per instantiation and LDS size (the occupancy query is a runtime call)
static int strip_resident_blocks_per_cu(bool pmj, int nw, bool ladder, size_t lds)
{
    static std::mutex mu;
    static std::vector<std::pair<uint64_t, int>> cache;
    const uint64_t key = (uint64_t(lds) << 8) | (uint64_t(pmj) << 2) | (uint64_t(ladder) << 1) | uint64_t(nw == 1);
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &e : cache)
        if (e.first == key) return e.second;
    const int n = strip_blocks_per_cu(pmj, nw, ladder, lds);
    cache.emplace_back(key, n);
    return n;
}

StripPlan strip_plan(const isingmc_states *s, size_t timesteps, bool ladder)
{
    StripPlan P;
    const isingmc_graph *g = s->g;
    const int mode = s->opt.strip;
    if (mode == 0 || s->strip_disabled || g->kind != ISINGMC_KIND_LATTICE2D || !g->vec || timesteps < 2) return P;
    const uint32_t qpr = g->geom.wpr / 4;
    if ((qpr & (qpr - 1)) != 0 || qpr > 32) return P; // power of two, at least two rows per wave
    P.nw = s->opt.strip_nw == 1 ? 1 : 4; // measured on 1024^2 x 64 (round 4, with wave priorities): 8.4 (4) / 8.6 (1) us per timestep; with exchange rounds 10.4 (4) / 12.0 (1)
    const uint32_t S = 64 * uint32_t(P.nw) / qpr;
    if (g->geom.H % S != 0 || g->geom.H / S < 2) return P;
    int dev_cus = 256;
    (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, g->device);
    // Every workgroup of a launch must be resident at once.  Per CU: what the runtime's occupancy calculation grants this
    // instantiation with its dynamic LDS (registers, LDS, wave slots), and never more than the policy bound of
    // STRIP_MAX_WAVES_PER_CU waves (beyond it the per-colour launches are faster anyway).
    const size_t lds = (size_t(2) * (S + 2) * g->geom.wpr + 16) * sizeof(uint32_t);
    const int by_occupancy = strip_resident_blocks_per_cu(!g->uniform_sign, P.nw, ladder, lds);
    const size_t per_cu = std::min<size_t>(size_t(STRIP_MAX_WAVES_PER_CU) / size_t(P.nw), size_t(std::max(by_occupancy, 0)));
    const size_t limit = per_cu * size_t(std::max(dev_cus, 1)); // workgroups resident at once
    const size_t n_strips = g->geom.H / S, total = s->R * n_strips;
    if (limit == 0 || n_strips > limit) return P;
    // one pass only by default: with twice the replicas (1024^2 x 128) the per-colour launches are long enough to win (16.7 vs 18.6 us)
    if (mode != 1 && total > size_t(s->opt.strip_max_wg >= 0 ? s->opt.strip_max_wg : int(limit))) return P;
    const size_t passes = (total + limit - 1) / limit;
    P.replicas_per_pass = (s->R + passes - 1) / passes;
    while (P.replicas_per_pass * n_strips > limit) P.replicas_per_pass--;
    if (P.replicas_per_pass == 0) return P;
    P.a.S = S;
    P.a.n_strips = uint32_t(n_strips);
    uint32_t ql = 0;
    while ((1u << ql) < qpr) ql++;
    P.a.qpr_log2 = ql;
    P.use = true;
    return P;
}

// strip launches of one process on one device never overlap: each needs all its workgroups resident at once
static std::mutex g_strip_mutex;
static hipEvent_t g_strip_done[64] = {};

// one pass: replicas [r0, r0 + n) for timesteps [s->t, s->t + nk).  steps_out / final_out: see lat_strip_kernel
int launch_strip(isingmc_states *s, const StripPlan &P, size_t r0, size_t n, size_t nk, const LatThr *d_thr_steps,
                 uint32_t thr_stride, unsigned long long *steps_out, double *final_energies, const StripLadder *ladder)
{
    const isingmc_graph *g = s->g;
    const size_t granules = s->cap * size_t(P.a.n_strips) * 4 * g->geom.wpr;
    if (s->halo_cap < granules) {
        HIP_TRY(stream_quiesce(s->stream)); // recycled blocks: nothing enqueued may still use the old one
        if (s->d_halo) HIP_TRY(cached_free(s->d_halo));
        s->d_halo = nullptr;
        s->halo_cap = 0;
        TRY(dev_alloc(&s->d_halo, granules));
        HIP_TRY(hipMemsetAsync(s->d_halo, 0, granules * sizeof(unsigned long long), s->stream));
        s->halo_cap = granules;
        s->strip_epoch = 0;
    }
    if (!s->d_strip_err) {
        TRY(dev_alloc(&s->d_strip_err, 4));
        HIP_TRY(hipMemsetAsync(s->d_strip_err, 0, 4 * sizeof(uint32_t), s->stream));
    }
    StripFinal fin{nullptr, nullptr, 0.0, 0};
    if (final_energies) {
        if (!s->d_strip_fin) {
            TRY(dev_alloc(&s->d_strip_fin, s->cap));
            HIP_TRY(hipMemsetAsync(s->d_strip_fin, 0, s->cap * sizeof(unsigned long long), s->stream));
        }
        fin = StripFinal{s->d_strip_fin + r0, final_energies + r0, g->jabs, 2ll * (long long)g->nvars};
    }
    if (uint64_t(s->strip_epoch) + 2 * nk + 2 >= 0xFFFFFFF0ull) { // tags are unique per states object: restart them
        HIP_TRY(hipMemsetAsync(s->d_halo, 0, s->halo_cap * sizeof(unsigned long long), s->stream));
        s->strip_epoch = 0;
    }
    // test hook (tests/test_gpu_strip.py): the first strip launch of this object runs with the error word already raised,
    // as if a workgroup had timed out -- in-order dispatch makes a real timeout need a co-tenant or a replica of more strips
    // than the chip holds -- so that the host's recovery (restore the planes, repeat on the per-colour launches) is exercised
    if (!s->strip_test_failed && s->opt.strip_test_fail_once) {
        const uint32_t one = STRIP_ERR_TIMEOUT;
        HIP_TRY(hipMemcpyAsync(s->d_strip_err, &one, sizeof one, hipMemcpyHostToDevice, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        s->strip_test_failed = true;
    }
    StripArgs a = P.a;
    a.epoch = s->strip_epoch;
    a.xcd_remap = n % 8 == 0;
    const size_t lds = (size_t(2) * (a.S + 2) * g->geom.wpr + 16) * sizeof(uint32_t);
    {
        std::lock_guard<std::mutex> lock(g_strip_mutex);
        hipEvent_t &ev = g_strip_done[g->device & 63];
        if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        else HIP_TRY(hipStreamWaitEvent(s->stream, ev, 0));
        HIP_TRY(strip_launch(!g->uniform_sign, P.nw, unsigned(n * a.n_strips), lds, s->stream, s->d_state + r0 * g->state_words, g->geom, a, s->t,
                             uint32_t(nk), s->d_keys + r0, d_thr_steps, thr_stride, s->has_betas ? s->d_thr + r0 : nullptr, g->d_jneg,
                             g->jneg_uniform, s->d_halo + r0 * size_t(a.n_strips) * 4 * g->geom.wpr, steps_out, fin,
                             ladder ? *ladder : StripLadder{}, uint32_t(s->R), s->d_strip_err));
        HIP_TRY(hipEventRecord(ev, s->stream));
    }
    return ISINGMC_OK;
}

// after a synchronisation: did a strip launch give up (its workgroups were not all resident)?  Then everything the strip
// kernels share between launches is reset and the object takes the per-colour launches from now on.  Callers that kept
// the planes they started from (run_steps, isingmc_run_sampling) repeat their work; the others report the error.
int strip_check(isingmc_states *s)
{
    if (!s->d_strip_err) return ISINGMC_OK;
    uint32_t h = 0;
    HIP_TRY(hipMemcpy(&h, s->d_strip_err, sizeof h, hipMemcpyDeviceToHost));
    if (h == 0) return ISINGMC_OK;
    (void)hipMemset(s->d_strip_err, 0, sizeof h);
    if (s->d_strip_fin) (void)hipMemset(s->d_strip_fin, 0, s->cap * sizeof(unsigned long long));
    if (s->d_pt_round_counts) (void)hipMemset(s->d_pt_round_counts, 0, 2 * s->R * sizeof(unsigned long long));
    if (s->d_pt_mail) (void)hipMemset(s->d_pt_mail, 0, 4 * s->R * sizeof(unsigned long long));
    if (s->d_halo) (void)hipMemset(s->d_halo, 0, s->halo_cap * sizeof(unsigned long long));
    s->strip_epoch = 0;
    s->meas_fresh = false;
    s->strip_disabled = true;
    return STRIP_TIMED_OUT;
}

int strip_error(int rc)
{
    if (rc != STRIP_TIMED_OUT) return rc;
    return fail(ISINGMC_ERR_HIP, "the persistent strip kernel timed out waiting for a neighbour strip (its workgroups were not all "
                                 "resident: is another process using this GPU?) inside a sequence of enqueue-only calls; the "
                                 "configurations of this object are invalid.  The object uses the per-colour launches from now on "
                                 "(ISINGMC_STRIP=0 selects them from the start)");
}

static int run_steps_impl(isingmc_states *s, size_t timesteps, const double *betas, size_t beta_stride, double *energies_per_step,
                          float *device_ms, bool sync, double *final_energies);

// planes of a states object before a call that may launch the strip kernel (D2D copy on the engine's stream: ~3 us for
// 1024^2 x 64), so that a timeout costs a repeat of the call instead of the configurations
int snapshot_take(isingmc_states *s)
{
    const size_t words = s->R * s->g->state_words;
    if (s->snapshot_cap < words) {
        HIP_TRY(stream_quiesce(s->stream));
        if (s->d_snapshot) HIP_TRY(cached_free(s->d_snapshot));
        s->d_snapshot = nullptr;
        s->snapshot_cap = 0;
        TRY(dev_alloc(&s->d_snapshot, words));
        s->snapshot_cap = words;
    }
    HIP_TRY(hipMemcpyAsync(s->d_snapshot, s->d_state, words * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
    return ISINGMC_OK;
}

int snapshot_restore(isingmc_states *s)
{
    HIP_TRY(hipMemcpyAsync(s->d_state, s->d_snapshot, s->R * s->g->state_words * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
    return ISINGMC_OK;
}

bool may_use_strips(const isingmc_states *s)
{
    return s && s->R && !s->packed && !s->strip_disabled && s->g->kind == ISINGMC_KIND_LATTICE2D && s->g->mc_mode == MC_NONE &&
           s->opt.strip != 0;
}

int run_steps(isingmc_states *s, size_t timesteps, const double *betas, size_t beta_stride,
              double *energies_per_step, float *device_ms, bool sync, double *final_energies)
{
    // a synchronous call keeps the planes it started from when it may launch the strip kernel; if a launch gives up (its
    // workgroups were not all resident: a co-tenant, a CU mask) the call is repeated with the per-colour launches
    const bool guard = sync && timesteps >= 2 && may_use_strips(s) && strip_plan(s, timesteps).use;
    if (!guard) {
        const int rc = run_steps_impl(s, timesteps, betas, beta_stride, energies_per_step, device_ms, sync, final_energies);
        return sync ? strip_error(rc) : rc;
    }
    TRY(use_device(s->g->device));
    const uint64_t t0 = s->t;
    TRY(snapshot_take(s));
    int rc = run_steps_impl(s, timesteps, betas, beta_stride, energies_per_step, device_ms, sync, final_energies);
    if (rc != STRIP_TIMED_OUT) return rc;
    TRY(snapshot_restore(s));
    s->t = t0;
    rc = run_steps_impl(s, timesteps, betas, beta_stride, energies_per_step, device_ms, sync, final_energies); // strip_disabled now
    return strip_error(rc);
}

static int run_steps_impl(isingmc_states *s, size_t timesteps, const double *betas, size_t beta_stride, double *energies_per_step,
                          float *device_ms, bool sync, double *final_energies)
{
    if (!s) return fail(ISINGMC_ERR_INVALID, "NULL states");
    if (timesteps && !betas && !s->has_betas) return fail(ISINGMC_ERR_INVALID, "betas is NULL");
    if (!s->has_betas)
        for (size_t k = 0; k < timesteps; k++)
            if (!std::isfinite(betas[k * beta_stride])) return fail(ISINGMC_ERR_INVALID, "beta must be finite");
    if (device_ms) *device_ms = 0.f;
    TRY(use_device(s->g->device));
    const isingmc_graph *g = s->g;
    const size_t R = s->R;
    if (R == 0) s->t += timesteps; // time passes for an empty container too (replicas appended later start here)
    if (R == 0 || timesteps == 0) return ISINGMC_OK;
    if (s->packed) return pk_run_steps(s, timesteps, betas, beta_stride, energies_per_step, device_ms, sync);
    const bool lattice = g->kind == ISINGMC_KIND_LATTICE2D;

    // per-step energies on the lattice path: integer counters per (step, replica), converted at the
    // end of each chunk; on the general path one measure() per step.
    // small lattices: one LDS-resident launch per chunk of timesteps instead of two launches per timestep
    // (up to 1024 quads per colour: beyond that one workgroup per replica is slower than the launches it saves)
    // lattices with a field or open boundaries: the multi-class kernels, one launch per colour (+ one measurement per step)
    const bool mc = lattice && g->mc_mode != MC_NONE;
    const bool resident = lattice && !mc && g->state_words * sizeof(uint32_t) <= LDS_RESIDENT_MAX_BYTES && g->geom.nquads <= 1024 &&
                          !resident_disabled(s);
    // per-step counters: 16 B per (step, replica) and counter slot, at most 32 MiB per chunk on each side of the bus
    const StripPlan strip = (lattice && !resident && !mc) ? strip_plan(s, timesteps) : StripPlan{};
    s->meas_fresh = false;
    const size_t step_slots = (energies_per_step && lattice && !resident && !strip.use && !mc) ? MEASURE_SLOTS : 1;
    size_t chunk = energies_per_step ? std::max<size_t>(1, std::min<size_t>(timesteps, (size_t(32) << 20) / (16 * R * step_slots))) : timesteps;
    const bool gen_resident = !lattice && gen_resident_fits(g, R) && !resident_disabled(s);
    // the multi-class modes' LDS-resident kernel: same size bound
    const bool mc_resident = mc && g->state_words * sizeof(uint32_t) <= LDS_RESIDENT_MAX_BYTES && g->geom.nquads <= 1024 && !resident_disabled(s);
    if (resident || gen_resident || strip.use || mc_resident) chunk = std::min<size_t>(chunk, 65536);
    DeviceScratch scratch(s->stream);
    double *d_beta_steps = nullptr, *d_gen_energies = nullptr;
    long long *d_gen_mags = nullptr;
    if (gen_resident) {
        if (!s->has_betas) TRY(scratch.alloc(&d_beta_steps, beta_stride ? chunk : 1));
        if (energies_per_step) TRY(scratch.alloc(&d_gen_energies, chunk * R));
    } else if (!lattice && energies_per_step) { // CSR path: one reduction slot per step, read back per chunk
        TRY(scratch.alloc(&d_gen_energies, chunk * R));
        TRY(scratch.alloc(&d_gen_mags, R));
    }
    unsigned long long *d_steps = nullptr;
    LatThr *d_thr_steps = nullptr;
    std::vector<unsigned long long> h_steps;
    std::vector<LatThr> h_thr;
    if (energies_per_step && lattice) {
        // streaming kernels measure inside the colour-1 half-sweep, into MEASURE_SLOTS partial counters per replica
        TRY(scratch.alloc(&d_steps, chunk * R * 2 * step_slots));
        h_steps.resize(chunk * R * 2 * step_slots);
    }
    if ((resident || strip.use) && !s->has_betas) TRY(scratch.alloc(&d_thr_steps, beta_stride ? chunk : 1));
    int rc = ISINGMC_OK;
    if (device_ms) HIP_TRY(hipEventRecord(s->ev0, s->stream));
    // mid-size launches (a few waves per SIMD) leave the GPU idle around every kernel boundary: run the
    // replica blocks on several streams.  Large launches (c2) keep the chip full on one stream.
    size_t want_lanes = 1;
    if (lattice && !resident && !strip.use && !mc_resident && !energies_per_step) { // the multi-class kernels' launches too
        const size_t waves_per_launch = R * ((g->geom.nquads + 255) / 256) * 4;
        if (s->opt.streams > 0) want_lanes = size_t(s->opt.streams);
        // < 64 waves per SIMD per launch: +17..33 % with 2 lanes (4 go host-bound); short calls lose it to fork/join.
        // Large launches: +2.8 % (one block's drain overlaps the other's ramp); the fork/join is ~45 us per call
        else if (waves_per_launch < 64 * 1024 ? timesteps >= 64 : timesteps >= 8) want_lanes = 2;
        want_lanes = std::min(want_lanes, R);
    }
    // every exit path below joins the lanes again: later calls (measure, get_states) use s->stream alone
    struct LaneJoin {
        isingmc_states *s;
        ~LaneJoin() { if (s->n_lanes > 1) (void)lanes_join(s); }
    } lane_join{s};
    if (want_lanes > 1) TRY(lanes_fork(s, want_lanes));
    for (size_t k0 = 0; k0 < timesteps && rc == ISINGMC_OK; k0 += chunk) {
        const size_t nk = std::min(chunk, timesteps - k0);
        if (d_steps) HIP_TRY(hipMemsetAsync(d_steps, 0, nk * R * 2 * step_slots * sizeof(unsigned long long), s->stream));
        if (resident) {
            if (!s->has_betas) {
                h_thr.resize(beta_stride ? nk : 1);
                for (size_t k = 0; k < h_thr.size(); k++) h_thr[k] = lattice_thresholds(betas[(k0 + k) * beta_stride], g->jabs);
                HIP_TRY(hipMemcpyAsync(d_thr_steps, h_thr.data(), h_thr.size() * sizeof(LatThr), hipMemcpyHostToDevice, s->stream));
            }
            // small lattices: eight (four, two) lanes per quad, one (two, four) Philox calls each (lat_resident_spread_kernel)
            // ... while every replica of the call is resident at once: beyond that the one-lane-per-quad kernel's small workgroups fill
            // the chip better (measured, tools/small_lattice_spread_ab.py: 64^2 x 2048 4.1 against 5.9 us, x 4096 10.5 against 9.1;
            // 128^2 x 512 4.4 against 5.4, x 1024 8.7 against 5.3).  ISINGMC_RESIDENT_SPREAD=0 / 2: never / whenever the lattice allows
            const int spread_mode = s->opt.resident_spread;
            // eight lanes per quad only: with four or two (256 / 512 quads per colour, 1024 threads) the barriers of a 16-wave workgroup
            // cost more than the shorter chain saves (256^2 x 64: 5.7 against 5.1 us; ISINGMC_RESIDENT_LPQ=4 / 2 for A/B runs)
            const int lpq_forced = s->opt.resident_lpq;
            const int lpq = lpq_forced == 4 || lpq_forced == 2 ? lpq_forced : 8;
            bool spread = spread_mode != 0 && size_t(g->geom.nquads) * size_t(lpq) <= 1024;
            const unsigned spread_threads = unsigned((size_t(g->geom.nquads) * size_t(lpq) + 63) / 64 * 64);
            const size_t spread_lds = g->state_words * sizeof(uint32_t) + size_t(g->geom.nquads) * 8 * sizeof(uint4);
            if (spread && spread_mode != 2) {
                int n_cu = 256;
                const int per_cu = spread_blocks_per_cu(g->vec, !g->uniform_sign, lpq, spread_threads, spread_lds);
                (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, g->device);
                spread = per_cu > 0 && R <= size_t(n_cu) * size_t(per_cu);
            }
            const unsigned threads = spread ? spread_threads : unsigned(std::min<size_t>(1024, (g->geom.nquads + 63) / 64 * 64));
            const size_t lds = spread ? spread_lds : g->state_words * sizeof(uint32_t);
            const auto launch = [&](auto kernel) {
                hipLaunchKernelGGL(kernel, dim3(unsigned(R)), dim3(threads), lds, s->stream,
                                   s->d_state, g->geom, s->t, uint32_t(nk), s->d_keys, d_thr_steps, uint32_t(beta_stride ? 1 : 0),
                                   s->has_betas ? s->d_thr : nullptr, g->d_jneg, g->jneg_uniform, d_steps, uint32_t(R));
            };
            if (spread) {
                HIP_TRY(spread_launch(g->vec, !g->uniform_sign, lpq, unsigned(R), threads, lds, s->stream, s->d_state, g->geom, s->t, uint32_t(nk), s->d_keys,
                                      d_thr_steps, uint32_t(beta_stride ? 1 : 0), s->has_betas ? s->d_thr : nullptr, g->d_jneg, g->jneg_uniform,
                                      d_steps, uint32_t(R)));
            } else if (g->vec) { if (g->uniform_sign) launch(lat_resident_kernel<true, false>); else launch(lat_resident_kernel<true, true>); }
            else { if (g->uniform_sign) launch(lat_resident_kernel<false, false>); else launch(lat_resident_kernel<false, true>); }
            s->t += nk;
            if (k0 + nk < timesteps && !d_steps) HIP_TRY(hipStreamSynchronize(s->stream)); // h_thr is reused by the next chunk
        }
        if (strip.use) { // mid-size lattices: the whole chunk of timesteps in one persistent launch per block of replicas
            if (!s->has_betas) {
                h_thr.resize(beta_stride ? nk : 1);
                for (size_t k = 0; k < h_thr.size(); k++) h_thr[k] = lattice_thresholds(betas[(k0 + k) * beta_stride], g->jabs);
                HIP_TRY(hipMemcpyAsync(d_thr_steps, h_thr.data(), h_thr.size() * sizeof(LatThr), hipMemcpyHostToDevice, s->stream));
            }
            // final_energies (device, [R]): the energies of the final configurations come with the last launch (tempering rounds)
            const bool last = k0 + nk == timesteps;
            for (size_t r0 = 0; r0 < R && rc == ISINGMC_OK; r0 += strip.replicas_per_pass)
                rc = launch_strip(s, strip, r0, std::min(strip.replicas_per_pass, R - r0), nk, d_thr_steps, uint32_t(beta_stride ? 1 : 0),
                                  d_steps ? d_steps + 2 * r0 : nullptr, last ? final_energies : nullptr);
            if (rc != ISINGMC_OK) break;
            s->strip_epoch += uint32_t(2 * nk);
            s->t += nk;
            if (final_energies && last) s->meas_fresh = true;
            if (k0 + nk < timesteps && !d_steps) HIP_TRY(hipStreamSynchronize(s->stream)); // h_thr is reused by the next chunk
        }
        if (mc_resident) {
            DeviceScratch thr_scratch(s->stream); // freed (after a stream sync) at the end of this chunk
            LatThrMC *d_thr_mc_steps = nullptr;
            if (!s->has_betas) {
                std::vector<LatThrMC> h(beta_stride ? nk : 1);
                for (size_t k = 0; k < h.size(); k++) h[k] = lattice_thresholds_mc(g, betas[(k0 + k) * beta_stride]);
                rc = thr_scratch.alloc(&d_thr_mc_steps, h.size());
                if (rc != ISINGMC_OK) break;
                HIP_TRY(hipMemcpy(d_thr_mc_steps, h.data(), h.size() * sizeof(LatThrMC), hipMemcpyHostToDevice));
            }
            // small lattices, few enough replicas to be resident at once: eight lanes per quad (lat_mc_resident_kernel SPREAD; the
            // kernel takes the spread form when the launch's LDS holds the random words too).  ISINGMC_RESIDENT_SPREAD=0: off
            const int mc_spread_mode = s->opt.resident_spread;
            int mc_n_cu = 256;
            (void)hipDeviceGetAttribute(&mc_n_cu, hipDeviceAttributeMultiprocessorCount, g->device);
            const size_t spread_threads = (size_t(g->geom.nquads) * 8 + 63) / 64 * 64;
            const bool mc_spread = mc_spread_mode != 0 && spread_threads <= 1024 &&
                                   (mc_spread_mode == 2 || R <= size_t(mc_n_cu) * std::max<size_t>(1, 1024 / spread_threads));
            const unsigned threads = mc_spread ? unsigned(spread_threads) : unsigned(std::min<size_t>(1024, (g->geom.nquads + 63) / 64 * 64));
            const size_t mc_lds = g->state_words * sizeof(uint32_t) + (mc_spread ? size_t(g->geom.nquads) * 8 * sizeof(uint4) : 0);
            for (size_t r0 = 0; r0 < R && rc == ISINGMC_OK; r0 += 65535) {
                const size_t n = std::min<size_t>(65535, R - r0);
                const hipError_t err = mc_launch_resident(g->mc_mode, !g->uniform_sign, unsigned(n), threads, mc_lds,
                                                          s->stream, s->d_state + r0 * g->state_words, g->geom, s->t, uint32_t(nk), s->d_keys + r0,
                                                          d_thr_mc_steps, uint32_t(beta_stride ? 1 : 0),
                                                          s->has_betas ? s->d_thr_mc + r0 : nullptr, g->d_jneg, g->jneg_uniform, g->open, g->d_fneg,
                                                          d_steps ? d_steps + 2 * r0 : nullptr, uint32_t(R));
                if (err != hipSuccess) rc = fail(ISINGMC_ERR_HIP, hipGetErrorString(err));
            }
            if (rc != ISINGMC_OK) break;
            s->t += nk;
        }
        if (gen_resident) {
            const size_t nb = beta_stride ? nk : 1;
            if (!s->has_betas) HIP_TRY(hipMemcpyAsync(d_beta_steps, betas + k0 * beta_stride, nb * sizeof(double), hipMemcpyHostToDevice, s->stream));
            unsigned threads = 64;
            for (uint32_t c = 0; c < g->n_colours; c++)
                threads = std::max<unsigned>(threads, unsigned(std::min<uint64_t>(1024, g->class_base[c + 1] - g->class_base[c])));
            // the graph in LDS too (gen_resident_kernel STAGE) while every replica of the call can still be resident at once (160 KB
            // of LDS per compute unit): small graphs, where a timestep is a chain of dependent loads.  Everything when that fits,
            // else the topology alone (the links of the chain); ISINGMC_GEN_STAGE=0 / 1 / 2 forces none / all / topology (A/B runs)
            const int stage_mode = s->opt.gen_stage;
            int n_cu = 256;
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, g->device);
            const size_t lds_cu = 160 * 1024, lds_max = 150 * 1024; // (a few KB stay free for the kernel's static LDS)
            size_t stage_bytes = 0;
            int stage = 0;
            {
                size_t bytes[3] = {0, 0, 0}, resident[3] = {0, 0, 0};
                for (int mode = 1; mode <= 2; mode++) {
                    bytes[mode] = size_t(gen_stage_words(g->gdev.n_pos, g->gen_edges2, g->gdev.bias != nullptr, g->w_is_float ? 4 : 8, mode)) * 4;
                    if (bytes[mode] <= lds_max) resident[mode] = std::min<size_t>(2048 / threads, lds_cu / (bytes[mode] + 1024)); // workgroups per compute unit
                }
                if (stage_mode == 0) stage = 0;
                else if (stage_mode == 1 || stage_mode == 2) stage = resident[stage_mode] ? stage_mode : 0;
                else if (resident[1] && (R <= size_t(n_cu) * resident[1] || threads > 512 || resident[2] <= resident[1])) stage = 1;
                // (measured, tools/small_graph_stage_ab.py: workgroups of <= 512 threads gain from running side by side, so when
                //  the full copy would keep some of the call's replicas waiting the smaller one wins: 32^2 x 1024 5.7 against
                //  6.2 us, 8^3 x 1024 3.5 against 4.4; 1024-thread workgroups do not: 12^3 x 512 6.6 against 9.4)
                else if (resident[2]) stage = 2;
                stage_bytes = bytes[stage];
            }
            const auto launch = [&](auto kernel) {
                if (stage_bytes > 64 * 1024) // beyond the default limit of dynamic LDS
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(stage_bytes));
                hipLaunchKernelGGL(kernel, dim3(unsigned(R)), dim3(threads), stage ? stage_bytes : g->state_words * sizeof(uint32_t), s->stream,
                                   s->d_state, g->gdev, s->t, uint32_t(nk), s->d_keys, d_beta_steps, uint32_t(beta_stride ? 1 : 0),
                                   s->has_betas ? s->d_beta : nullptr, d_gen_energies, g->self_energy, g->gen_edges2);
            };
            if (g->w_is_float) {
                if (stage == 1) launch(gen_resident_kernel<float, 1>); else if (stage == 2) launch(gen_resident_kernel<float, 2>); else launch(gen_resident_kernel<float, 0>);
            } else {
                if (stage == 1) launch(gen_resident_kernel<double, 1>); else if (stage == 2) launch(gen_resident_kernel<double, 2>); else launch(gen_resident_kernel<double, 0>);
            }
            s->t += nk;
            if (d_gen_energies) {
                std::vector<double> he(nk * R);
                hipError_t err = hipMemcpyAsync(he.data(), d_gen_energies, he.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream);
                if (err == hipSuccess) err = hipStreamSynchronize(s->stream);
                if (err != hipSuccess) { rc = fail(ISINGMC_ERR_HIP, hipGetErrorString(err)); break; }
                for (size_t r = 0; r < R; r++)
                    for (size_t k = 0; k < nk; k++) energies_per_step[r * timesteps + k0 + k] = he[r * nk + k];
            } else if (k0 + nk < timesteps) {
                HIP_TRY(hipStreamSynchronize(s->stream));
            }
        }
        for (size_t k = k0; k < k0 + nk && !resident && !gen_resident && !strip.use && !mc_resident; k++) {
            const double beta = s->has_betas ? 0.0 : betas[k * beta_stride];
            if (mc) {
                const LatThrMC thr = lattice_thresholds_mc(g, beta);
                const size_t per_lane = (R + s->n_lanes - 1) / s->n_lanes; // replica blocks on the lanes' streams, as launch_lat_sweep
                for (uint32_t colour = 0; colour < 2 && rc == ISINGMC_OK; colour++)
                    for (size_t lane = 0; lane < s->n_lanes && rc == ISINGMC_OK; lane++) {
                        const size_t lo = lane * per_lane, hi = std::min(R, lo + per_lane);
                        hipStream_t stream = s->n_lanes > 1 ? s->lanes[lane] : s->stream;
                        for (size_t r0 = lo; r0 < hi; r0 += MAX_GRID_Y) {
                            const size_t n = std::min(MAX_GRID_Y, hi - r0);
                            const hipError_t err = mc_launch_sweep(g->mc_mode, !g->uniform_sign, lat_grid(g, g->geom.nquads, n), stream,
                                                                   s->d_state + r0 * g->state_words, g->geom, colour, s->t, s->d_keys + r0, thr,
                                                                   s->has_betas ? s->d_thr_mc + r0 : nullptr, g->d_jneg, g->jneg_uniform, g->open,
                                                                   g->d_fneg);
                            if (err != hipSuccess) { rc = fail(ISINGMC_ERR_HIP, hipGetErrorString(err)); break; }
                        }
                    }
                if (rc != ISINGMC_OK) break;
                if (d_steps) { // get_energy after this timestep: a measurement pass behind the sweep (as on the general path)
                    s->t++;
                    rc = measure_enqueue(s, d_steps + (k - k0) * R * 2, nullptr, nullptr);
                    if (rc != ISINGMC_OK) break;
                    continue;
                }
            } else if (lattice) {
                const LatThr thr = lattice_thresholds(beta, g->jabs);
                LAT_DISPATCH(launch_lat_sweep, s, 0u, thr, s->t);
                if (d_steps) LAT_DISPATCH(launch_lat_sweep_measure, s, thr, s->t, d_steps + (k - k0) * R * 2 * step_slots, 2 * step_slots);
                else LAT_DISPATCH(launch_lat_sweep, s, 1u, thr, s->t);
            } else {
                launch_gen_timestep(s, beta);
            }
            s->t++;
            if (energies_per_step && !lattice) {
                rc = measure_enqueue(s, nullptr, d_gen_energies + (k - k0) * R, d_gen_mags);
                if (rc != ISINGMC_OK) break;
            }
        }
        if (energies_per_step && !lattice && !gen_resident && rc == ISINGMC_OK) {
            std::vector<double> he(nk * R);
            hipError_t err = hipMemcpyAsync(he.data(), d_gen_energies, he.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream);
            if (err == hipSuccess) err = hipStreamSynchronize(s->stream);
            if (err != hipSuccess) { rc = fail(ISINGMC_ERR_HIP, hipGetErrorString(err)); break; }
            for (size_t k = 0; k < nk; k++)
                for (size_t r = 0; r < R; r++) energies_per_step[r * timesteps + k0 + k] = he[k * R + r] + g->self_energy;
        }
        if (d_steps && rc == ISINGMC_OK) {
            hipError_t err = hipMemcpyAsync(h_steps.data(), d_steps, nk * R * 2 * step_slots * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream);
            if (err == hipSuccess) err = hipStreamSynchronize(s->stream);
            if (err != hipSuccess) { rc = fail(ISINGMC_ERR_HIP, hipGetErrorString(err)); break; }
            for (size_t k = 0; k < nk; k++)
                for (size_t r = 0; r < R; r++) {
                    unsigned long long sat = 0, up = 0;
                    for (size_t sl = 0; sl < step_slots; sl++) {
                        sat += h_steps[((k * R + r) * step_slots + sl) * 2];
                        up += h_steps[((k * R + r) * step_slots + sl) * 2 + 1];
                    }
                    energies_per_step[r * timesteps + k0 + k] = lattice_energy(g, sat, up);
                }
        }
    }
    if (s->n_lanes > 1) { const int jrc = lanes_join(s); if (rc == ISINGMC_OK) rc = jrc; }
    if (device_ms && rc == ISINGMC_OK) {
        hipError_t err = hipEventRecord(s->ev1, s->stream);
        if (err == hipSuccess) err = hipEventSynchronize(s->ev1);
        if (err == hipSuccess) err = hipEventElapsedTime(device_ms, s->ev0, s->ev1);
        if (err != hipSuccess) rc = fail(ISINGMC_ERR_HIP, hipGetErrorString(err));
    }
    if (rc != ISINGMC_OK) return rc;
    HIP_TRY(hipGetLastError());
    if (sync) {
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (strip.use) TRY(strip_check(s));
    }
    return ISINGMC_OK;
}

extern "C" int isingmc_do_time_steps(isingmc_states *s, size_t timesteps, const double *betas, size_t beta_stride,
                                     double *energies_per_step)
{
    return run_steps(s, timesteps, betas, beta_stride, energies_per_step, nullptr);
}

extern "C" int isingmc_do_time_steps_timed(isingmc_states *s, size_t timesteps, const double *betas,
                                           size_t beta_stride, float *device_ms_out)
{
    if (!device_ms_out) return fail(ISINGMC_ERR_INVALID, "device_ms_out is NULL");
    return run_steps(s, timesteps, betas, beta_stride, nullptr, device_ms_out);
}

extern "C" int isingmc_get_energies(isingmc_states *s, double *energies_out)
{
    if (!s || !energies_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    TRY(use_device(s->g->device));
    return measure(s, energies_out, nullptr);
}

extern "C" int isingmc_get_magnetisations(isingmc_states *s, int64_t *mags_out)
{
    if (!s || !mags_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    TRY(use_device(s->g->device));
    return measure(s, nullptr, mags_out);
}

extern "C" int isingmc_get_packed_states(isingmc_states *s, uint32_t *words_out)
{
    if (!s || !words_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    TRY(use_device(s->g->device));
    if (s->R == 0) return ISINGMC_OK;
    if (s->packed) return pk_get_states(s, nullptr, 0, words_out);
    HIP_TRY(hipMemcpyAsync(words_out, s->d_state, s->R * s->g->state_words * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return ISINGMC_OK;
}

// ------------------------------------------------------------------------------------------------
// measurements enqueued behind the sweeps (per-step energies, sampling, tempering rounds)
// ------------------------------------------------------------------------------------------------
int measure_enqueue(isingmc_states *s, unsigned long long *counts_slot, double *e_slot, long long *m_slot, bool want_up)
{
    const isingmc_graph *g = s->g;
    const size_t R = s->R;
    if (s->packed) { // counts_slot: [pk_slots()][2], one pair per (group, bit) -- a shard may own only some bits of a group
        HIP_TRY(hipMemsetAsync(counts_slot, 0, 2 * s->pk_slots() * sizeof(unsigned long long), s->stream));
        if (s->rj) {
            const bool bip = g->n_colours == 2;
            int dev_cus = 256;
            (void)hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, g->device);
            // all workgroups resident at once (the runtime's occupancy figure for this instantiation), every one walks its
            // share of the blocks; a grid one round and a bit long would run its tail at a fraction of the chip
            static std::mutex per_cu_mutex; // (the device fan-out measures from several host threads)
            static int per_cu[6][2][2] = {};
            int pc;
            {
                std::lock_guard<std::mutex> lock(per_cu_mutex);
                int &slot = per_cu[g->rj.slots == 4 ? 0 : g->rj.slots == 7 ? 1 : g->rj.slots == 11 ? 2 : g->rj.slots == 15 ? 3 : g->rj.slots == 23 ? 4 : 5][bip][want_up];
                if (slot == 0) slot = std::max(1, rj_measure_blocks_per_cu(g->rj.slots, bip, want_up));
                pc = slot;
            }
            const size_t resident = size_t(pc) * size_t(std::max(dev_cus, 1));
            // two colour classes: the bonds from class 0 alone; class 1 is visited only for its bias terms or the up spins
            const uint32_t class0_end = bip ? uint32_t(g->class_base[1]) : 0u;
            const uint32_t scan_end = bip && !g->has_bias && !want_up ? class0_end : g->pk.n_pos;
            const size_t scan_blocks = scan_end / rj_threads(g->rj.slots);
            for (size_t g0 = 0; g0 < s->groups; g0 += MAX_GRID_Y) {
                const size_t ng = std::min(MAX_GRID_Y, s->groups - g0);
                const size_t gx = std::min(scan_blocks, std::max<size_t>(1, resident / ng));
                // hi level (+ the up spins when wanted) into the first counter of a slot, lo level into the second
                HIP_TRY(rj_launch_measure(dim3(unsigned(std::max<size_t>(gx, 1)), unsigned(ng)), s->stream, s->d_state + g0 * g->pk.n_pos, g->rj_hi,
                                          g->pk.site, class0_end, scan_end, want_up, counts_slot + 2 * 32 * g0));
                if (!want_up)
                    HIP_TRY(rj_launch_measure(dim3(unsigned(std::max<size_t>(gx, 1)), unsigned(ng)), s->stream, s->d_state + g0 * g->pk.n_pos, g->rj_lo,
                                              g->pk.site, class0_end, scan_end, false, counts_slot + 2 * 32 * g0 + 1));
            }
            return ISINGMC_OK;
        }
        uint32_t ppt = PK_MEASURE_POS_PER_THREAD; // halved until the launch has >= 1024 workgroups (not below 8: the transpose
                                                  // at the end of a chunk costs as much as ~16 positions)
        while (ppt > 8 && size_t((g->pk.n_pos + 256 * ppt - 1) / (256 * ppt)) * s->groups < 1024) ppt /= 2;
        const unsigned blocks = unsigned(std::max<uint32_t>(1, std::min<uint32_t>(2048, (g->pk.n_pos + 256 * ppt - 1) / (256 * ppt))));
        for (size_t g0 = 0; g0 < s->groups; g0 += MAX_GRID_Y) {
            const size_t ng = std::min(MAX_GRID_Y, s->groups - g0);
            hipLaunchKernelGGL(pk_measure_kernel, dim3(blocks, unsigned(ng)), dim3(256), 0, s->stream,
                               s->d_state + g0 * g->pk.n_pos, g->pk, counts_slot + 2 * 32 * g0, uint32_t(32 * ng), ppt,
                               g->n_colours == 2 ? uint32_t(g->class_base[1]) : g->pk.n_pos, g->n_colours == 2 ? 2u : 1u);
        }
    } else if (g->kind == ISINGMC_KIND_LATTICE2D) {
        HIP_TRY(hipMemsetAsync(counts_slot, 0, 2 * R * sizeof(unsigned long long), s->stream));
        LAT_DISPATCH(launch_lat_measure, s, counts_slot, size_t(2));
    } else {
        for (size_t r0 = 0; r0 < R; r0 += MAX_GRID_Y) {
            const size_t n = std::min(MAX_GRID_Y, R - r0);
            const dim3 grid(s->n_partials, unsigned(n));
            if (g->w_is_float)
                hipLaunchKernelGGL(gen_measure_kernel<float>, grid, dim3(256), 0, s->stream, s->d_state + r0 * g->state_words,
                                   g->gdev, s->d_pe + r0 * s->n_partials, s->d_pm + r0 * s->n_partials);
            else
                hipLaunchKernelGGL(gen_measure_kernel<double>, grid, dim3(256), 0, s->stream, s->d_state + r0 * g->state_words,
                                   g->gdev, s->d_pe + r0 * s->n_partials, s->d_pm + r0 * s->n_partials);
        }
        hipLaunchKernelGGL(gen_reduce_kernel, dim3(unsigned(R)), dim3(256), 0, s->stream, s->d_pe, s->d_pm, s->n_partials,
                           e_slot, m_slot);
    }
    HIP_TRY(hipGetLastError());
    return ISINGMC_OK;
}

void lat_measure_enqueue(isingmc_states *s, unsigned long long *out, size_t out_stride)
{
    LAT_DISPATCH(launch_lat_measure, s, out, out_stride);
}
