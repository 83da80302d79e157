// libisingmc.so: configurations out of the device -- isingmc_get_states and the sampling run (lattice.rs:231-299), both as a
// two-slab pipeline: device sweeps, a copy stream, and the host's bit -> bool expansion overlap.  No kernel is launched here.
#include "internal.hpp"

static int sampling_reserve(isingmc_states *s, size_t words, size_t counts, size_t energies);
static void madvise_hugepages(void *p, size_t bytes);

extern "C" int isingmc_get_states(isingmc_states *s, uint8_t *states_out, size_t replica_stride_bytes)
{
    if (!s || !states_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    if (replica_stride_bytes < s->g->nvars) return fail(ISINGMC_ERR_INVALID, "replica stride smaller than nvars");
    TRY(use_device(s->g->device));
    const isingmc_graph *g = s->g;
    if (s->packed) return s->R ? pk_get_states(s, states_out, replica_stride_bytes, nullptr) : ISINGMC_OK;
    // packed device words -> pinned host memory in slabs of replicas (<= 64 MiB), two in flight on the copy stream, expanded
    // to bytes by the host threads while the next slab crosses PCIe (the buffers of the sampling pipeline)
    if (s->R == 0) return ISINGMC_OK;
    const size_t slab = std::max<size_t>(1, std::min<size_t>(s->R, (size_t(64) << 20) / (g->state_words * 4)));
    const size_t n_slabs = (s->R + slab - 1) / slab;
    TRY(sampling_reserve(s, slab * g->state_words, 0, 0));
    madvise_hugepages(states_out, s->R * replica_stride_bytes);
    HIP_TRY(hipEventRecord(s->sample_ready[0], s->stream)); // everything queued on the engine's stream comes first
    HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->sample_ready[0], 0));
    const auto copy_slab = [&](size_t j) {
        const size_t r0 = j * slab, n = std::min(slab, s->R - r0);
        HIP_TRY(hipMemcpyAsync(s->h_samples[j & 1], s->d_state + r0 * g->state_words, n * g->state_words * sizeof(uint32_t),
                               hipMemcpyDeviceToHost, s->copy_stream));
        HIP_TRY(hipEventRecord(s->sample_copied[j & 1], s->copy_stream));
        return int(ISINGMC_OK);
    };
    TRY(copy_slab(0));
    for (size_t j = 0; j < n_slabs; j++) {
        HIP_TRY(hipEventSynchronize(s->sample_copied[j & 1]));
        if (j + 1 < n_slabs) TRY(copy_slab(j + 1)); // into the other buffer, which slab j-1's expansion has released
        const size_t r0 = j * slab, n = std::min(slab, s->R - r0);
        const uint32_t *words = s->h_samples[j & 1];
        parallel_for(n, [&](size_t i) { unpack_state(g, words + i * g->state_words, states_out + (r0 + i) * replica_stride_bytes); }, g->nvars);
    }
    return ISINGMC_OK;
}

// ------------------------------------------------------------------------------------------------
// sampling run: thermalise, then S x { freq timesteps; record state + energy }  (lattice.rs:271-287,
// classicising.rs:144-173).  Everything is enqueued on the stream -- sweeps, a device-to-device copy of
// the packed configurations into a sample ring, the measurement kernels -- and the host only waits once
// per chunk of samples (<= 512 MiB of packed states), then expands the bits to bools on its threads.
// ------------------------------------------------------------------------------------------------
// buffers of the sampling pipeline, grown on demand and kept for the next call (pinning host memory is slow)
template <typename T>
static int regrow(T **dev, T **host, size_t count)
{
    if (*dev) HIP_TRY(cached_free(*dev));
    if (*host) HIP_TRY(cached_host_free(*host));
    *dev = nullptr;
    *host = nullptr;
    TRY(dev_alloc(dev, count));
    HIP_TRY(cached_host_malloc(reinterpret_cast<void **>(host), std::max<size_t>(count, 1) * sizeof(T)));
    return ISINGMC_OK;
}

static int sampling_reserve(isingmc_states *s, size_t words, size_t counts, size_t energies)
{
    if (!s->copy_stream) HIP_TRY(pooled_stream_create(&s->copy_stream));
    HIP_TRY(stream_quiesce(s->stream)); // buffers regrown below go through the block caches
    HIP_TRY(stream_quiesce(s->copy_stream));
    for (int b = 0; b < 2; b++) {
        if (!s->sample_ready[b]) HIP_TRY(pooled_event_create(&s->sample_ready[b], true));
        if (!s->sample_copied[b]) HIP_TRY(pooled_event_create(&s->sample_copied[b], true));
    }
    HIP_TRY(hipStreamSynchronize(s->copy_stream));
    if (words > s->sample_cap_words) {
        s->sample_cap_words = 0;
        for (int b = 0; b < 2; b++) TRY(regrow(&s->d_samples[b], &s->h_samples[b], words));
        s->sample_cap_words = words;
    }
    if (counts > s->sample_cap_counts) {
        s->sample_cap_counts = 0;
        for (int b = 0; b < 2; b++) TRY(regrow(&s->d_sample_counts[b], &s->h_counts[b], counts));
        s->sample_cap_counts = counts;
    }
    if (energies > s->sample_cap_e) {
        s->sample_cap_e = 0;
        for (int b = 0; b < 2; b++) TRY(regrow(&s->d_sample_e[b], &s->h_e[b], energies));
        if (s->d_sample_m) HIP_TRY(cached_free(s->d_sample_m));
        s->d_sample_m = nullptr;
        TRY(dev_alloc(&s->d_sample_m, energies));
        s->sample_cap_e = energies;
    }
    return ISINGMC_OK;
}

// large output arrays are touched for the first time by the expansion threads: with transparent huge pages the first
// touch costs one fault per 2 MiB instead of one per 4 KiB (a hint; ignored where THP is off)
static void madvise_hugepages(void *p, size_t bytes)
{
    if (bytes < (size_t(32) << 20)) return;
    const uintptr_t lo = (reinterpret_cast<uintptr_t>(p) + 0x1FFFFF) & ~uintptr_t(0x1FFFFF);
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(p) + bytes) & ~uintptr_t(0x1FFFFF);
    if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
}

static int run_sampling_impl(isingmc_states *s, double beta, size_t thermalization, size_t sampling_freq, size_t n_samples,
                             double *energies_out, uint8_t *states_out);

extern "C" int isingmc_run_sampling(isingmc_states *s, double beta, size_t thermalization, size_t sampling_freq,
                                    size_t n_samples, double *energies_out, uint8_t *states_out)
{
    if (!s) return fail(ISINGMC_ERR_INVALID, "NULL states");
    if (!may_use_strips(s) || !(strip_plan(s, thermalization).use || strip_plan(s, sampling_freq).use))
        return strip_error(run_sampling_impl(s, beta, thermalization, sampling_freq, n_samples, energies_out, states_out));
    // as run_steps: the call keeps the planes it started from and is repeated without the strip kernel if a launch gives up
    TRY(use_device(s->g->device));
    const uint64_t t0 = s->t;
    TRY(snapshot_take(s));
    int rc = run_sampling_impl(s, beta, thermalization, sampling_freq, n_samples, energies_out, states_out);
    if (rc != STRIP_TIMED_OUT) return rc;
    TRY(snapshot_restore(s));
    s->t = t0;
    return strip_error(run_sampling_impl(s, beta, thermalization, sampling_freq, n_samples, energies_out, states_out));
}

static int run_sampling_impl(isingmc_states *s, double beta, size_t thermalization, size_t sampling_freq, size_t n_samples,
                             double *energies_out, uint8_t *states_out)
{
    if (!s) return fail(ISINGMC_ERR_INVALID, "NULL states");
    if (sampling_freq == 0) return fail(ISINGMC_ERR_INVALID, "sampling_freq must be positive");
    if (n_samples && s->R && (!energies_out || !states_out)) return fail(ISINGMC_ERR_INVALID, "NULL output");
    if (!s->has_betas && !std::isfinite(beta)) return fail(ISINGMC_ERR_INVALID, "beta must be finite");
    TRY(use_device(s->g->device));
    const isingmc_graph *g = s->g;
    const size_t R = s->R, S = n_samples, N = g->nvars;
    // a uniform beta is installed as per-replica thresholds for the duration of the call: the step
    // launches then need no per-call host tables and nothing in the loop synchronises
    struct BetaGuard {
        isingmc_states *s;
        bool active;
        ~BetaGuard() { if (active) (void)isingmc_states_set_betas(s, nullptr); }
    } guard{s, false};
    if (!s->has_betas && R) {
        const std::vector<double> b(R, beta);
        TRY(set_betas(s, b.data(), /*all_equal=*/true));
        guard.active = true;
    }
    TRY(run_steps(s, thermalization, nullptr, 0, nullptr, nullptr, /*sync=*/false));
    if (R == 0 || S == 0) {
        if (R == 0) s->t += S * sampling_freq;
        HIP_TRY(hipStreamSynchronize(s->stream));
        return ISINGMC_OK;
    }
    const bool counts = s->packed || g->kind == ISINGMC_KIND_LATTICE2D;
    const size_t words = s->packed ? s->groups * size_t(g->pk.n_pos) : R * g->state_words;
    const size_t CS = s->packed ? s->pk_slots() : R; // counter pairs per sample
    // Pipeline over SLABS of samples (<= 64 MiB of packed words each), two in flight: while the host expands
    // slab j-1 from pinned memory into the caller's bool[R,S,N] array (non-temporal stores, all host threads), the device
    // runs the sweeps of slab j and a second stream copies finished slabs out.  The expansion to one byte per spin is the
    // floor of this call (the reference's output format: 8x the packed bytes, host memory bandwidth); the sweeps, the
    // sample copies and PCIe hide behind it, or it hides behind them when sampling_freq is large.
    const size_t slab_bytes = size_t(std::max(1, s->opt.sample_slab_bytes)); // (tests shrink it)
    const size_t slab = std::max<size_t>(1, std::min<size_t>(S, slab_bytes / (words * sizeof(uint32_t))));
    const size_t n_slabs = (S + slab - 1) / slab;
    TRY(sampling_reserve(s, slab * words, counts ? slab * CS * 2 : 0, counts ? 0 : slab * R));
    madvise_hugepages(states_out, R * S * N);
    const auto unpack_slab = [&](size_t j) {
        const size_t k0 = j * slab, nk = std::min(slab, S - k0), b = j & 1;
        const uint32_t *h_samples = s->h_samples[b];
        const unsigned long long *h_counts = s->h_counts[b];
        const double *h_e = s->h_e[b];
        parallel_for(nk * R, [&](size_t idx) {
            const size_t k = idx / R, r = idx % R;
            uint8_t *out = states_out + (r * S + k0 + k) * N;
            double energy;
            if (s->packed) {
                const size_t sl = r + s->pk_bit0;
                const uint32_t *w = h_samples + k * words + (sl / 32) * g->pk.n_pos;
                const uint32_t bit = uint32_t(sl % 32);
                for (uint64_t i = 0; i < N; i++) out[i] = (w[g->pos[i]] >> bit) & 1u;
                energy = pk_energy(g, s->rj, h_counts[(k * CS + sl) * 2], h_counts[(k * CS + sl) * 2 + 1]);
            } else {
                unpack_state(g, h_samples + k * words + r * g->state_words, out);
                if (counts) energy = lattice_energy(g, h_counts[(k * R + r) * 2], h_counts[(k * R + r) * 2 + 1]);
                else energy = h_e[k * R + r] + g->self_energy;
            }
            energies_out[r * S + k0 + k] = energy;
        }, N);
    };
    for (size_t j = 0; j < n_slabs; j++) {
        const size_t k0 = j * slab, nk = std::min(slab, S - k0), b = j & 1;
        // device slab b is free: its copy-out (slab j-2) was awaited before slab j-2 was expanded, in iteration j-1
        for (size_t k = 0; k < nk; k++) {
            TRY(run_steps(s, sampling_freq, nullptr, 0, nullptr, nullptr, /*sync=*/false));
            HIP_TRY(hipMemcpyAsync(s->d_samples[b] + k * words, s->d_state, words * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
            TRY(measure_enqueue(s, counts ? s->d_sample_counts[b] + k * CS * 2 : nullptr, counts ? nullptr : s->d_sample_e[b] + k * R,
                                counts ? nullptr : s->d_sample_m, /*want_up=*/false));
        }
        HIP_TRY(hipEventRecord(s->sample_ready[b], s->stream));
        HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->sample_ready[b], 0));
        HIP_TRY(hipMemcpyAsync(s->h_samples[b], s->d_samples[b], nk * words * sizeof(uint32_t), hipMemcpyDeviceToHost, s->copy_stream));
        if (counts) HIP_TRY(hipMemcpyAsync(s->h_counts[b], s->d_sample_counts[b], nk * CS * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->copy_stream));
        else HIP_TRY(hipMemcpyAsync(s->h_e[b], s->d_sample_e[b], nk * R * sizeof(double), hipMemcpyDeviceToHost, s->copy_stream));
        HIP_TRY(hipEventRecord(s->sample_copied[b], s->copy_stream));
        if (j >= 1) { // expand the previous slab while the device works on this one
            HIP_TRY(hipEventSynchronize(s->sample_copied[1 - b]));
            unpack_slab(j - 1);
        }
    }
    HIP_TRY(hipEventSynchronize(s->sample_copied[(n_slabs - 1) & 1]));
    unpack_slab(n_slabs - 1);
    HIP_TRY(hipStreamSynchronize(s->stream));
    TRY(strip_check(s));
    return ISINGMC_OK;
}

