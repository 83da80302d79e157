// libisingmc.so: on-stream parallel tempering (isingmc_pt_*) and the in-process ladder across several devices
// (isingmc_pt_group_*: RCCL resolved with dlopen).
#include "internal.hpp"

// ------------------------------------------------------------------------------------------------
// on-stream parallel tempering (no host synchronisation inside the sweep / measure / swap loop)
// ------------------------------------------------------------------------------------------------
extern "C" int isingmc_states_stream(isingmc_states *s, void **stream_out)
{
    if (!s || !stream_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    *stream_out = s->stream;
    return ISINGMC_OK;
}

extern "C" int isingmc_synchronize(isingmc_states *s)
{
    if (!s) return fail(ISINGMC_ERR_INVALID, "NULL states");
    TRY(use_device(s->g->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return strip_error(strip_check(s));
}

// where the exchange kernel writes a local slot's acceptance data: {T3, T4} per replica on the lattice path, the RjBeta of
// the slot's bit position on the real-coupling path
static uint64_t *pt_thr_local(isingmc_states *s)
{
    if (s->packed && s->rj) return reinterpret_cast<uint64_t *>(s->d_rj_betas + s->pk_bit0);
    if (s->packed) return reinterpret_cast<uint64_t *>(s->d_pk_slot_thr); // pk_bit0 == 0 (checked at attach)
    return reinterpret_cast<uint64_t *>(s->d_thr);
}

// bit-sliced packed path: the groups' threshold tables follow the slots' new thresholds (enqueue only)
static int pt_after_swap(isingmc_states *s)
{
    if (s->packed && !s->rj && s->R) {
        hipLaunchKernelGGL(pk_tables_from_slots_kernel, dim3(unsigned(s->groups)), dim3(64), 0, s->stream, s->d_pk_slot_thr, uint32_t(s->R), s->d_tab);
        HIP_TRY(hipGetLastError());
    }
    return ISINGMC_OK;
}

// why this container cannot take an on-stream ladder of that geometry ("" when it can): no side effects
static std::string pt_attach_obstacle(const isingmc_states *s, size_t n_rungs, size_t slot_offset, size_t slots_per_rank, size_t world_size)
{
    if (s->pt_attached) return "a ladder is already attached";
    const bool pk_ladder = s->packed && !s->rj;
    if (!s->packed && (s->g->kind != ISINGMC_KIND_LATTICE2D || s->g->mc_mode != MC_NONE))
        return "on-stream tempering is implemented for periodic, field-free lattices and for the replica-packed "
               "paths (use the host swap step)";
    // the replicas of a bit-sliced group number their ties together: a shard must hold whole groups (distributed.block_size aligns them)
    if (pk_ladder && (s->pk_bit0 != 0 || ((s->first + s->R) % 32 != 0 && s->first + s->R != s->n_total)))
        return "a tempering shard on the replica-packed path must start and end on multiples of 32 slots";
    if (slot_offset + s->R > n_rungs || s->R > slots_per_rank || slots_per_rank * world_size < n_rungs || n_rungs >= 0xFFFFFFFFull)
        return "ladder / shard geometry mismatch";
    return "";
}

extern "C" int isingmc_pt_can_attach(const isingmc_states *s, size_t n_rungs, size_t slot_offset, size_t slots_per_rank,
                                     size_t world_size, int *ok_out)
{
    if (!s || !ok_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    const std::string why = pt_attach_obstacle(s, n_rungs, slot_offset, slots_per_rank, world_size);
    *ok_out = why.empty() ? 1 : 0;
    if (!why.empty()) (void)fail(ISINGMC_OK, why); // (informational: the call itself succeeded)
    return ISINGMC_OK;
}

// the ladder's device buffers go back; the container keeps its configurations and timestep and takes uniform betas again
extern "C" int isingmc_pt_detach(isingmc_states *s)
{
    if (!s) return fail(ISINGMC_ERR_INVALID, "NULL states");
    if (!s->pt_attached) return ISINGMC_OK;
    TRY(use_device(s->g->device));
    HIP_TRY(hipStreamSynchronize(s->stream)); // enqueued rounds may still read the ladder
    TRY(strip_error(strip_check(s)));
    for (void **p : {(void **)&s->d_pt_ladder, (void **)&s->d_pt_local, (void **)&s->d_pt_all, (void **)&s->d_pt_ladder_thr,
                     (void **)&s->d_pt_perm, (void **)&s->d_pt_counters, (void **)&s->d_pt_mail, (void **)&s->d_pt_round_counts,
                     (void **)&s->d_pt_perm2}) {
        if (*p) (void)cached_free(*p);
        *p = nullptr;
    }
    s->pt = PtDev{};
    s->pt_attached = false;
    s->has_betas = false;
    s->betas.clear();
    s->meas_fresh = false;
    return ISINGMC_OK;
}

extern "C" int isingmc_pt_attach(isingmc_states *s, const double *ladder_betas, size_t n_rungs, size_t slot_offset,
                                 size_t slots_per_rank, size_t world_size, uint64_t seed)
{
    if (!s || !ladder_betas) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    {
        const std::string why = pt_attach_obstacle(s, n_rungs, slot_offset, slots_per_rank, world_size);
        if (!why.empty()) return fail(ISINGMC_ERR_INVALID, why);
    }
    const bool rj_ladder = s->packed && s->rj, pk_ladder = s->packed && !s->rj;
    for (size_t i = 0; i < n_rungs; i++)
        if (!std::isfinite(ladder_betas[i])) return fail(ISINGMC_ERR_INVALID, "beta must be finite");
    TRY(use_device(s->g->device));
    const isingmc_graph *g = s->g;
    const bool lattice = g->kind == ISINGMC_KIND_LATTICE2D;
    TRY(dev_alloc(&s->d_pt_ladder, n_rungs));
    TRY(dev_alloc(&s->d_pt_perm, n_rungs));
    TRY(dev_alloc(&s->d_pt_local, slots_per_rank));
    TRY(dev_alloc(&s->d_pt_all, slots_per_rank * world_size));
    TRY(dev_alloc(&s->d_pt_counters, 2));
    std::vector<uint32_t> perm(n_rungs);
    for (size_t i = 0; i < n_rungs; i++) perm[i] = uint32_t(i);
    HIP_TRY(hipMemcpy(s->d_pt_ladder, ladder_betas, n_rungs * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_pt_perm, perm.data(), n_rungs * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(s->d_pt_counters, 0, 2 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(s->d_pt_local, 0, slots_per_rank * sizeof(double)));
    HIP_TRY(hipMemset(s->d_pt_all, 0, slots_per_rank * world_size * sizeof(double)));
    if (rj_ladder) { // acceptance scales per rung (host arithmetic: the bits of isingmc_states_set_betas)
        std::vector<uint64_t> thr(n_rungs);
        for (size_t i = 0; i < n_rungs; i++) {
            RjBeta b;
            rj_beta(ladder_betas[i], g->rj_k, &b.shift, &b.mant);
            std::memcpy(&thr[i], &b, sizeof b);
        }
        TRY(dev_alloc(&s->d_pt_ladder_thr, n_rungs));
        HIP_TRY(hipMemcpy(s->d_pt_ladder_thr, thr.data(), thr.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        if (!s->d_rj_betas) TRY(dev_alloc(&s->d_rj_betas, 32 * s->groups));
        std::vector<RjBeta> init(32 * s->groups, RjBeta{31u, 0xFFFFFFFFu}); // bits this shard does not own: accept-all, nobody reads them
        HIP_TRY(hipMemcpy(s->d_rj_betas, init.data(), init.size() * sizeof(RjBeta), hipMemcpyHostToDevice));
    }
    if (pk_ladder) { // T_m per rung, m = 1 .. PK_MAX_DEG: the values pk_fill_table puts into the host-built tables
        std::vector<uint64_t> thr(size_t(PK_MAX_DEG) * n_rungs);
        for (size_t i = 0; i < n_rungs; i++)
            for (uint32_t m = 1; m <= uint32_t(PK_MAX_DEG); m++) thr[i * PK_MAX_DEG + m - 1] = threshold_fixed(ladder_betas[i], 2.0 * g->jabs * double(m));
        TRY(dev_alloc(&s->d_pt_ladder_thr, thr.size()));
        HIP_TRY(hipMemcpy(s->d_pt_ladder_thr, thr.data(), thr.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        if (!s->d_pk_slot_thr) TRY(dev_alloc(&s->d_pk_slot_thr, size_t(32) * s->groups * PK_MAX_DEG));
        HIP_TRY(hipMemset(s->d_pk_slot_thr, 0, size_t(32) * s->groups * PK_MAX_DEG * sizeof(unsigned long long)));
        if (!s->d_tab) TRY(dev_alloc(&s->d_tab, s->groups * PK_TAB_WORDS));
    }
    if (lattice) { // thresholds per rung from the host's exp: bit-identical to isingmc_states_set_betas
        std::vector<uint64_t> thr(2 * n_rungs);
        for (size_t i = 0; i < n_rungs; i++) {
            const LatThr t = lattice_thresholds(ladder_betas[i], g->jabs);
            thr[2 * i] = t.T3;
            thr[2 * i + 1] = t.T4;
        }
        TRY(dev_alloc(&s->d_pt_ladder_thr, 2 * n_rungs));
        HIP_TRY(hipMemcpy(s->d_pt_ladder_thr, thr.data(), thr.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    s->pt = PtDev{s->d_pt_ladder, s->d_pt_ladder_thr, rj_ladder ? 1u : pk_ladder ? uint32_t(PK_MAX_DEG) : 2u, s->d_pt_perm, s->d_pt_all, s->d_pt_counters, uint32_t(n_rungs),
                  uint32_t(slot_offset), uint32_t(s->R), uint32_t(seed), uint32_t(seed >> 32)};
    s->pt_per = slots_per_rank;
    s->pt_world = world_size;
    s->betas.assign(s->R, 0.0);
    s->has_betas = true;
    s->pt_attached = true;
    hipLaunchKernelGGL(pt_swap_kernel, dim3(1), dim3(1024), 0, s->stream, s->pt, pt_thr_local(s), s->packed ? nullptr : s->d_beta, 1u);
    HIP_TRY(hipGetLastError());
    TRY(pt_after_swap(s));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return ISINGMC_OK;
}

extern "C" int isingmc_pt_buffers(isingmc_states *s, void **d_local_out, void **d_all_out, size_t *per_rank_out)
{
    if (!s || !s->pt_attached) return fail(ISINGMC_ERR_INVALID, "no ladder attached");
    if (d_local_out) *d_local_out = s->d_pt_local;
    if (d_all_out) *d_all_out = s->d_pt_all;
    if (per_rank_out) *per_rank_out = s->pt_per;
    return ISINGMC_OK;
}

extern "C" int isingmc_pt_time_steps(isingmc_states *s, size_t timesteps)
{
    if (!s || !s->pt_attached) return fail(ISINGMC_ERR_INVALID, "no ladder attached");
    // the strip kernel measures the final configurations itself: isingmc_pt_measure then needs no pass over the planes
    return run_steps(s, timesteps, nullptr, 0, nullptr, nullptr, /*sync=*/false,
                     /*final_energies=*/s->pt_world == 1 ? s->d_pt_all + s->pt.slot_offset : s->d_pt_local);
}

// The loop of tempering.rs:177-194 { timesteps(swap_every); parallel_tempering_step } for `timesteps` sweeps in ONE library
// call (enqueue only), with an exchange round after every swap_every-th sweep.  Single rank + strip geometry: one persistent
// launch whose strips exchange temperatures pair by pair through rung-indexed mailboxes (StripLadder), only the last
// round at a kernel boundary; otherwise the per-round sequence of the calls above.  Ranks > 1 must interleave their
// all-gather and therefore keep calling isingmc_pt_time_steps / _measure / _swap themselves.
extern "C" int isingmc_pt_run(isingmc_states *s, size_t timesteps, size_t swap_every)
{
    if (!s || !s->pt_attached) return fail(ISINGMC_ERR_INVALID, "no ladder attached");
    if (swap_every == 0) return fail(ISINGMC_ERR_INVALID, "swap_every must be positive");
    if (s->pt_world != 1) return fail(ISINGMC_ERR_INVALID, "isingmc_pt_run is for a single rank: the all-gather of a sharded ladder sits between measure and swap");
    TRY(use_device(s->g->device));
    const isingmc_graph *g = s->g;
    const size_t rounds = timesteps / swap_every, tail = timesteps % swap_every;
    const StripPlan P = s->R ? strip_plan(s, rounds * swap_every, /*ladder=*/true) : StripPlan{};
    const bool in_kernel = P.use && P.replicas_per_pass >= s->R && rounds >= 2 && rounds * swap_every <= 65536 &&
                           s->R == s->pt.n_rungs && s->opt.pt_in_kernel != 0;
    if (in_kernel) {
        const size_t R = s->R, nk = rounds * swap_every;
        if (!s->d_pt_mail) {
            TRY(dev_alloc(&s->d_pt_mail, 4 * R));
            TRY(dev_alloc(&s->d_pt_round_counts, 2 * R));
            TRY(dev_alloc(&s->d_pt_perm2, R));
            HIP_TRY(hipMemsetAsync(s->d_pt_mail, 0, 4 * R * sizeof(unsigned long long), s->stream));
            HIP_TRY(hipMemsetAsync(s->d_pt_round_counts, 0, 2 * R * sizeof(unsigned long long), s->stream));
        }
        // the number of the first round is on the device (exchange rounds never synchronise with the host): read it once here
        unsigned long long c[2];
        HIP_TRY(hipStreamSynchronize(s->stream));
        HIP_TRY(hipMemcpy(c, s->d_pt_counters, sizeof c, hipMemcpyDeviceToHost));
        const StripLadder lad{s->d_pt_ladder, reinterpret_cast<const unsigned long long *>(s->d_pt_ladder_thr), s->d_pt_perm, s->d_pt_perm2,
                              s->d_pt_mail, s->d_pt_round_counts, s->d_pt_counters, c[0], uint32_t(R), uint32_t(swap_every), s->pt.seed_lo,
                              s->pt.seed_hi, g->jabs, 2ll * (long long)g->nvars};
        s->meas_fresh = false;
        TRY(launch_strip(s, P, 0, R, nk, nullptr, 0, nullptr, s->d_pt_all + s->pt.slot_offset, &lad));
        s->strip_epoch += uint32_t(2 * nk);
        s->t += nk;
        s->meas_fresh = true; // the launch wrote the final energies
        HIP_TRY(hipMemcpyAsync(s->d_pt_perm, s->d_pt_perm2, R * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
        TRY(isingmc_pt_measure(s));
        TRY(isingmc_pt_swap(s)); // the last round of the block, at the kernel boundary (it also relabels d_thr / d_beta)
    } else {
        for (size_t k = 0; k < rounds; k++) {
            TRY(isingmc_pt_time_steps(s, swap_every));
            TRY(isingmc_pt_measure(s));
            TRY(isingmc_pt_swap(s));
        }
    }
    if (tail) TRY(isingmc_pt_time_steps(s, tail));
    return ISINGMC_OK;
}

// enqueue: energies of the local slots -> the local send buffer (and straight into the gathered
// buffer when there is a single rank)
extern "C" int isingmc_pt_measure(isingmc_states *s)
{
    if (!s || !s->pt_attached) return fail(ISINGMC_ERR_INVALID, "no ladder attached");
    TRY(use_device(s->g->device));
    const isingmc_graph *g = s->g;
    const size_t R = s->R;
    if (R == 0) return ISINGMC_OK;
    if (g->kind == ISINGMC_KIND_LATTICE2D) {
        // three launches per round: the conversion kernel leaves the counters zeroed for the next round and, on
        // a single rank, writes straight into the gathered array (no memset, no device-to-device copy)
        if (s->meas_fresh) { // the last strip launch of isingmc_pt_time_steps has already written these energies
            s->meas_fresh = false;
            return ISINGMC_OK;
        }
        if (!s->meas_zero) HIP_TRY(hipMemsetAsync(s->d_meas, 0, 2 * R * sizeof(unsigned long long), s->stream));
        lat_measure_enqueue(s, s->d_meas, size_t(2));
        hipLaunchKernelGGL(lat_energy_from_counts_kernel, dim3(unsigned((R + 255) / 256)), dim3(256), 0, s->stream, s->d_meas,
                           uint32_t(R), g->jabs, 2ll * (long long)g->nvars,
                           s->pt_world == 1 ? s->d_pt_all + s->pt.slot_offset : s->d_pt_local);
        s->meas_zero = true;
    } else if (s->packed && !s->rj) {
        TRY(measure_enqueue(s, s->d_meas, nullptr, nullptr, /*want_up=*/false));
        s->meas_zero = false;
        hipLaunchKernelGGL(pk_energy_from_counts_kernel, dim3(unsigned((R + 255) / 256)), dim3(256), 0, s->stream, s->d_meas, uint32_t(s->pk_bit0),
                           uint32_t(R), g->jabs, double(int64_t(g->n_directed / 2)), g->self_energy,
                           s->pt_world == 1 ? s->d_pt_all + s->pt.slot_offset : s->d_pt_local);
    } else if (s->packed && s->rj) {
        TRY(measure_enqueue(s, s->d_meas, nullptr, nullptr, /*want_up=*/false));
        s->meas_zero = false;
        HIP_TRY(rj_launch_energy_from_counts(s->stream, s->d_meas, uint32_t(s->pk_bit0), uint32_t(R), g->rj_k_energy, g->self_energy,
                                             s->pt_world == 1 ? s->d_pt_all + s->pt.slot_offset : s->d_pt_local));
    } else {
        return fail(ISINGMC_ERR_INVALID, "on-stream tempering is implemented for the lattice path and the real-coupling path; use the host swap step");
    }
    HIP_TRY(hipGetLastError());
    return ISINGMC_OK;
}

// enqueue: one exchange round from the gathered energies; relabels the local slots
extern "C" int isingmc_pt_swap(isingmc_states *s)
{
    if (!s || !s->pt_attached) return fail(ISINGMC_ERR_INVALID, "no ladder attached");
    TRY(use_device(s->g->device));
    hipLaunchKernelGGL(pt_swap_kernel, dim3(1), dim3(1024), 0, s->stream, s->pt, pt_thr_local(s), s->packed ? nullptr : s->d_beta, 0u);
    HIP_TRY(hipGetLastError());
    return pt_after_swap(s);
}

// synchronises; perm_out: uint32[n_rungs] (rung -> slot)
extern "C" int isingmc_pt_state(isingmc_states *s, uint32_t *perm_out, uint64_t *round_out, uint64_t *swaps_out)
{
    if (!s || !s->pt_attached) return fail(ISINGMC_ERR_INVALID, "no ladder attached");
    TRY(use_device(s->g->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    TRY(strip_error(strip_check(s)));
    unsigned long long c[2];
    HIP_TRY(hipMemcpy(c, s->d_pt_counters, sizeof c, hipMemcpyDeviceToHost));
    if (perm_out) HIP_TRY(hipMemcpy(perm_out, s->d_pt_perm, s->pt.n_rungs * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (round_out) *round_out = c[0];
    if (swaps_out) *swaps_out = c[1];
    return ISINGMC_OK;
}

// ------------------------------------------------------------------------------------------------
// In-process ladder across several devices: the shards of ONE beta ladder, each an isingmc_states on its
// own device with the ladder attached (isingmc_pt_attach, world_size = the number of shards), driven from ONE host thread.
// Between measure and swap every shard needs every shard's energies -- the swap step of tempering.rs:191-194 -- :
//   * RCCL: ncclAllGather of the `local` buffers into the `all` buffers, one communicator per shard (ncclCommInitAll), enqueued on
//     each shard's engine stream inside ncclGroupStart / ncclGroupEnd.  librccl.so is resolved with dlopen at group creation:
//     a single-GPU user of libisingmc.so needs no RCCL at all, and a Rust / pyo3 host needs no NCCL binding of its own.
//   * device copies (the shards share a device, RCCL is not installed, or ISINGMC_PT_GROUP_BACKEND=copy): the same gather as
//     events + hipMemcpyPeerAsync on the engines' streams.
// Both leave the same bytes in every `all` buffer; nothing in the loop waits on the host.
// ------------------------------------------------------------------------------------------------
#include <dlfcn.h>

namespace {
struct Rccl {
    void *handle = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok() const { return handle && CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd; }
};
constexpr int RCCL_FLOAT64 = 8; // ncclFloat64 (rccl.h)

Rccl &rccl()
{
    static Rccl *r = [] {
        auto *x = new Rccl;
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            x->handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x->handle) break;
        }
        if (x->handle) {
            x->CommInitAll = reinterpret_cast<decltype(x->CommInitAll)>(dlsym(x->handle, "ncclCommInitAll"));
            x->CommDestroy = reinterpret_cast<decltype(x->CommDestroy)>(dlsym(x->handle, "ncclCommDestroy"));
            x->AllGather = reinterpret_cast<decltype(x->AllGather)>(dlsym(x->handle, "ncclAllGather"));
            x->GroupStart = reinterpret_cast<decltype(x->GroupStart)>(dlsym(x->handle, "ncclGroupStart"));
            x->GroupEnd = reinterpret_cast<decltype(x->GroupEnd)>(dlsym(x->handle, "ncclGroupEnd"));
            x->GetErrorString = reinterpret_cast<decltype(x->GetErrorString)>(dlsym(x->handle, "ncclGetErrorString"));
        }
        return x;
    }();
    return *r;
}
} // namespace

struct isingmc_pt_group {
    std::vector<isingmc_states *> shards;
    std::vector<void *> comms;        // RCCL communicators, one per shard (empty: device copies)
    std::vector<hipEvent_t> measured; // per shard: its energies are in its `local` buffer
    size_t per = 0;

    ~isingmc_pt_group()
    {
        for (size_t k = 0; k < shards.size(); k++) {
            (void)hipSetDevice(shards[k]->g->device);
            (void)hipStreamSynchronize(shards[k]->stream);
            if (k < comms.size() && comms[k]) (void)rccl().CommDestroy(comms[k]);
            if (k < measured.size() && measured[k]) (void)hipEventDestroy(measured[k]);
        }
    }
};

static int rccl_fail(int rc, const char *what)
{
    const char *msg = rccl().GetErrorString ? rccl().GetErrorString(rc) : "?";
    return fail(ISINGMC_ERR_HIP, std::string(what) + ": " + msg);
}

extern "C" int isingmc_pt_group_create(isingmc_states **shards, size_t n_shards, int backend, isingmc_pt_group **group_out)
{
    if (!shards || !group_out || n_shards == 0) return fail(ISINGMC_ERR_INVALID, "NULL argument / no shards");
    *group_out = nullptr;
    size_t offset = 0;
    std::vector<int> devices;
    bool distinct = true;
    for (size_t k = 0; k < n_shards; k++) {
        const isingmc_states *s = shards[k];
        if (!s || !s->pt_attached) return fail(ISINGMC_ERR_INVALID, "every shard needs an attached ladder (isingmc_pt_attach)");
        if (s->pt_world != n_shards || s->pt_per != shards[0]->pt_per || s->pt.n_rungs != shards[0]->pt.n_rungs ||
            s->pt.seed_lo != shards[0]->pt.seed_lo || s->pt.seed_hi != shards[0]->pt.seed_hi)
            return fail(ISINGMC_ERR_INVALID, "the shards are not attached to one ladder (world size, slots per rank, rungs, seed)");
        if (s->pt.slot_offset != offset || s->pt.slot_offset != k * s->pt_per)
            return fail(ISINGMC_ERR_INVALID, "shard k must own the slots from k * slots_per_rank on, in rank order");
        offset += s->R;
        for (int d : devices) distinct &= d != s->g->device;
        devices.push_back(s->g->device);
    }
    if (offset != shards[0]->pt.n_rungs) return fail(ISINGMC_ERR_INVALID, "the shards do not cover the ladder");
    auto grp = std::make_unique<isingmc_pt_group>();
    grp->shards.assign(shards, shards + n_shards);
    grp->per = shards[0]->pt_per;
    grp->measured.assign(n_shards, nullptr);
    for (size_t k = 0; k < n_shards; k++) {
        TRY(use_device(devices[k]));
        HIP_TRY(hipEventCreateWithFlags(&grp->measured[k], hipEventDisableTiming));
    }
    // backend: 0 = RCCL when the devices are distinct (or there is one shard) and librccl.so resolves, else device copies;
    // 1 = RCCL or fail; 2 = device copies
    const char *env = std::getenv("ISINGMC_PT_GROUP_BACKEND");
    if (backend == 0 && env) backend = std::string(env) == "rccl" ? 1 : std::string(env) == "copy" ? 2 : 0;
    const bool want_rccl = backend == 1 || (backend == 0 && distinct && rccl().ok());
    if (backend == 1 && !rccl().ok()) return fail(ISINGMC_ERR_HIP, "librccl.so could not be loaded (dlopen)");
    if (backend == 1 && !distinct) return fail(ISINGMC_ERR_INVALID, "RCCL needs one device per shard (ncclCommInitAll refuses duplicates)");
    if (want_rccl) {
        grp->comms.assign(n_shards, nullptr);
        const int rc = rccl().CommInitAll(grp->comms.data(), int(n_shards), devices.data());
        if (rc != 0) {
            grp->comms.clear();
            return rccl_fail(rc, "ncclCommInitAll");
        }
    }
    *group_out = grp.release();
    return ISINGMC_OK;
}

extern "C" int isingmc_pt_group_backend(const isingmc_pt_group *grp)
{
    return grp && !grp->comms.empty() ? 1 : 2;
}

// enqueue: every shard's `local` energies -> every shard's `all` buffer (rank-major, slots_per_rank each)
extern "C" int isingmc_pt_group_allgather(isingmc_pt_group *grp)
{
    if (!grp) return fail(ISINGMC_ERR_INVALID, "NULL group");
    const size_t n = grp->shards.size();
    if (!grp->comms.empty()) {
        int rc = rccl().GroupStart();
        if (rc != 0) return rccl_fail(rc, "ncclGroupStart");
        for (size_t k = 0; k < n && rc == 0; k++) {
            isingmc_states *s = grp->shards[k];
            rc = rccl().AllGather(s->d_pt_local, s->d_pt_all, grp->per, RCCL_FLOAT64, grp->comms[k], s->stream);
        }
        const int rc2 = rccl().GroupEnd();
        if (rc != 0) return rccl_fail(rc, "ncclAllGather");
        if (rc2 != 0) return rccl_fail(rc2, "ncclGroupEnd");
        return ISINGMC_OK;
    }
    for (size_t j = 0; j < n; j++) { // shard j's measurement is complete ...
        TRY(use_device(grp->shards[j]->g->device));
        HIP_TRY(hipEventRecord(grp->measured[j], grp->shards[j]->stream));
    }
    for (size_t k = 0; k < n; k++) { // ... before any shard k copies it
        isingmc_states *s = grp->shards[k];
        TRY(use_device(s->g->device));
        for (size_t j = 0; j < n; j++) {
            const isingmc_states *src = grp->shards[j];
            if (j != k) HIP_TRY(hipStreamWaitEvent(s->stream, grp->measured[j], 0));
            HIP_TRY(hipMemcpyPeerAsync(s->d_pt_all + j * grp->per, s->g->device, src->d_pt_local, src->g->device, grp->per * sizeof(double), s->stream));
        }
    }
    // a shard's `local` buffer is overwritten by its next measurement: that must wait for the copies the OTHER shards made of it
    for (size_t k = 0; k < n; k++) {
        TRY(use_device(grp->shards[k]->g->device));
        HIP_TRY(hipEventRecord(grp->measured[k], grp->shards[k]->stream));
    }
    for (size_t j = 0; j < n; j++) {
        TRY(use_device(grp->shards[j]->g->device));
        for (size_t k = 0; k < n; k++)
            if (k != j) HIP_TRY(hipStreamWaitEvent(grp->shards[j]->stream, grp->measured[k], 0));
    }
    return ISINGMC_OK;
}

// enqueue: the loop of tempering.rs:177-194 over the whole ladder -- `timesteps` sweeps on every shard with an exchange round
// (measure, all-gather, swap) after every swap_every-th one
extern "C" int isingmc_pt_group_run(isingmc_pt_group *grp, size_t timesteps, size_t swap_every)
{
    if (!grp) return fail(ISINGMC_ERR_INVALID, "NULL group");
    if (swap_every == 0) return fail(ISINGMC_ERR_INVALID, "swap_every must be positive");
    for (size_t done = 0; done < timesteps;) {
        const size_t b = std::min(swap_every, timesteps - done);
        for (isingmc_states *s : grp->shards) TRY(isingmc_pt_time_steps(s, b));
        done += b;
        if (b == swap_every) {
            for (isingmc_states *s : grp->shards) TRY(isingmc_pt_measure(s));
            TRY(isingmc_pt_group_allgather(grp));
            for (isingmc_states *s : grp->shards) TRY(isingmc_pt_swap(s));
        }
    }
    return ISINGMC_OK;
}

extern "C" int isingmc_pt_group_synchronize(isingmc_pt_group *grp)
{
    if (!grp) return fail(ISINGMC_ERR_INVALID, "NULL group");
    for (isingmc_states *s : grp->shards) TRY(isingmc_synchronize(s));
    return ISINGMC_OK;
}

extern "C" void isingmc_pt_group_destroy(isingmc_pt_group *grp) { delete grp; }
