// Internal declarations shared by the host translation units of libisingmc.so (not installed, not part of the C ABI):
//   core.hip       error state, device-block / pinned-block / stream / event caches, environment helpers
//   graph.hip      edge-list checks, host-only C ABI helpers, lattice / general / packed / real-coupling graph construction
//   isingmc.hip    replica containers, every sweep / measurement launch, the persistent strip kernel's host side
//   sampling.hip   get_states and the double-buffered sampling pipeline
//   tempering.hip  on-stream parallel tempering, the in-process ladder group (RCCL through dlopen)
//   debug.hip      shader-clock probe
#pragma once
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/isingmc.h"
#include "general_kernels.hpp"
#include "host_logic.hpp"
#include "lattice_kernels.hpp"
#include "packed_kernels.hpp"
#include "mc_types.hpp"
#include "real_types.hpp"
#include "spread_types.hpp"
#include "strip_types.hpp"

using namespace isingmc;


#define IM_INTERNAL __attribute__((visibility("hidden")))

// ---- errors (core.hip) ----------------------------------------------------------------------------------------------------
IM_INTERNAL int fail(int code, const std::string &msg); // sets the calling thread's isingmc_last_error() and returns `code`
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t err__ = (expr);                                                                 \
        if (err__ != hipSuccess)                                                                   \
            return fail(err__ == hipErrorOutOfMemory ? ISINGMC_ERR_ALLOC : ISINGMC_ERR_HIP,        \
                        std::string(#expr) + ": " + hipGetErrorString(err__));                     \
    } while (0)

#define TRY(expr)                                                                                  \
    do {                                                                                           \
        int rc__ = (expr);                                                                         \
        if (rc__ != ISINGMC_OK) return rc__;                                                       \
    } while (0)


// ---- caches (core.hip): device blocks, pinned host blocks, streams, events are recycled; every owner drains its streams before
// it hands a block back (hipFree used to synchronise the device)
IM_INTERNAL hipError_t cached_malloc(void **out, size_t bytes);
IM_INTERNAL hipError_t cached_free(void *p);
IM_INTERNAL hipError_t cached_host_malloc(void **out, size_t bytes);
IM_INTERNAL hipError_t cached_host_free(void *p);
IM_INTERNAL hipError_t pooled_stream_create(hipStream_t *out);
IM_INTERNAL hipError_t stream_quiesce(hipStream_t st); // a query when everything has completed, else a synchronisation
IM_INTERNAL void pooled_stream_destroy(hipStream_t st);
IM_INTERNAL hipError_t pooled_event_create(hipEvent_t *out, bool disable_timing);
IM_INTERNAL void pooled_event_destroy(hipEvent_t ev, bool disable_timing);
IM_INTERNAL int use_device(int device);
IM_INTERNAL bool env_flag(const char *name);
IM_INTERNAL int env_int(const char *name, int dflt);

// ------------------------------------------------------------------------------------------------
// Path / tuning switches.  They are read from the environment ONCE, when a graph or a replica container is created, into an
// options block of that handle; isingmc_states_set_option changes one of them on one container.  Nothing below *_create calls
// getenv: two containers of one process can run different paths, and a test that sets a variable must do so before it creates
// the object it wants to steer.  (Measurement / A-B switches and test hooks; none of them changes a result, only which kernel
// produces it -- except FORCE / DISABLE of a kernel FAMILY, which select another spec: DESIGN.md section 2.)
// ------------------------------------------------------------------------------------------------
struct Options {
    int force_real = 0, disable_real = 0, force_packed = 0, disable_packed = 0; // kernel family (decided at creation)
    int disable_packed_uniform = 0, disable_resident = 0;
    int strip = -1, strip_nw = 4, strip_max_wg = -1, strip_test_fail_once = 0; // persistent strip kernel: -1 auto / 0 off / 1 force
    int streams = 0, pk_streams = 0;                                            // replica lanes: 0 = the measured rule
    int sweep_iters = 0, debug_sweep_lds = 0;
    int resident_spread = 1, resident_lpq = 0, gen_stage = -1;
    int sample_slab_bytes = 64 << 20, pt_in_kernel = 1, real_target_wgs = 3072;

    static Options from_env()
    {
        Options o;
        o.force_real = env_flag("ISINGMC_FORCE_REAL");
        o.disable_real = env_flag("ISINGMC_DISABLE_REAL");
        o.force_packed = env_flag("ISINGMC_FORCE_PACKED");
        o.disable_packed = env_flag("ISINGMC_DISABLE_PACKED");
        o.disable_packed_uniform = env_flag("ISINGMC_DISABLE_PACKED_UNIFORM");
        o.disable_resident = env_flag("ISINGMC_DISABLE_RESIDENT");
        o.strip = env_int("ISINGMC_STRIP", -1);
        o.strip_nw = env_int("ISINGMC_STRIP_NW", 4);
        o.strip_max_wg = env_int("ISINGMC_STRIP_MAX_WG", -1);
        o.strip_test_fail_once = env_flag("ISINGMC_STRIP_TEST_FAIL_ONCE");
        o.streams = env_int("ISINGMC_STREAMS", 0);
        o.pk_streams = env_int("ISINGMC_PK_STREAMS", 0);
        o.sweep_iters = env_int("ISINGMC_SWEEP_ITERS", 0);
        o.debug_sweep_lds = env_int("ISINGMC_DEBUG_SWEEP_LDS", 0);
        o.resident_spread = env_int("ISINGMC_RESIDENT_SPREAD", 1);
        o.resident_lpq = env_int("ISINGMC_RESIDENT_LPQ", 0);
        o.gen_stage = env_int("ISINGMC_GEN_STAGE", -1);
        o.sample_slab_bytes = std::max(1, env_int("ISINGMC_SAMPLE_SLAB_BYTES", 64 << 20));
        o.pt_in_kernel = env_int("ISINGMC_PT_IN_KERNEL", 1);
        o.real_target_wgs = std::max(256, env_int("ISINGMC_REAL_TARGET_WGS", 3072));
        return o;
    }

    // name: the environment variable's name without the ISINGMC_ prefix, lower or upper case
    bool set(const std::string &name_in, long value)
    {
        std::string n;
        for (char c : name_in) n += char(std::tolower(static_cast<unsigned char>(c)));
        if (n.rfind("isingmc_", 0) == 0) n = n.substr(8);
        const std::pair<const char *, int *> table[] = {
            {"force_real", &force_real}, {"disable_real", &disable_real}, {"force_packed", &force_packed}, {"disable_packed", &disable_packed},
            {"disable_packed_uniform", &disable_packed_uniform}, {"disable_resident", &disable_resident}, {"strip", &strip},
            {"strip_nw", &strip_nw}, {"strip_max_wg", &strip_max_wg}, {"strip_test_fail_once", &strip_test_fail_once},
            {"streams", &streams}, {"pk_streams", &pk_streams}, {"sweep_iters", &sweep_iters}, {"debug_sweep_lds", &debug_sweep_lds},
            {"resident_spread", &resident_spread}, {"resident_lpq", &resident_lpq}, {"gen_stage", &gen_stage},
            {"sample_slab_bytes", &sample_slab_bytes}, {"pt_in_kernel", &pt_in_kernel}, {"real_target_wgs", &real_target_wgs}};
        for (const auto &e : table)
            if (n == e.first) { *e.second = int(value); return true; }
        return false;
    }
};

struct isingmc_graph {
    int device = 0;
    int kind = ISINGMC_KIND_GENERAL;
    uint64_t nvars = 0, n_edges = 0;
    uint64_t state_words = 0;
    bool has_bias = false;
    // lattice path
    LatGeom geom{};
    bool vec = false;
    double jabs = 0.0;
    bool uniform_sign = true;
    uint32_t jneg_uniform = 0;
    uint32_t *d_jneg = nullptr; // [2 colours][4 directions][wpp]
    // multi-class checkerboard kernels (mc_types.hpp): uniform field or open boundaries on a recognised lattice
    int mc_mode = MC_NONE;
    double jabs_y = 0.0;  // MC_ANISO: |J| of the vertical bonds (jabs = the horizontal ones')
    double field = 0.0;   // MC_FIELD: h of E = sum J s s - h sum s
    McOpen open{0, 0, 0}; // MC_OPEN, MC_FIELD_OPEN
    uint32_t *d_fneg = nullptr; // fields of one size and both signs: sign planes [2][wpp] (bit set where h_i < 0); field = |h| then
    // general path
    GenGraphDev gdev{};
    uint32_t gen_edges2 = 0; // directed edges of the CSR (rowptr[n_pos])
    bool w_is_float = false;
    std::vector<uint64_t> class_base;
    std::vector<uint64_t> pos; // site -> packed position
    double self_energy = 0.0;
    uint32_t n_colours = 2;
    // replica-packed variant of the general path (uniform |J|, no fields, degree <= PK_MAX_DEG)
    bool packed_ok = false;
    PkGraphDev pk{};
    // every real site has this degree (3..6): packed_uni_kernels.hpp; 0 otherwise
    int pk_uni_deg = 0;
    bool pk_uni_pmj = false;             // couplings of both signs
    PkUniHeaders pk_uni{};
    uint32_t pk_uni_but_one = 0;         // (block, slot) headers that are a translation for every lane but one
    std::vector<uint32_t> pk_class_full; // per colour class: end of its last 256-block without padding
    std::vector<uint8_t> pk_class_table; // per colour class: some block header of its full blocks is PK_HDR_MIXED (needs table entries)
    uint64_t n_directed = 0;
    // replica-packed real-coupling path (real_kernels.hpp): any couplings and biases, degree <= 15
    bool rj_ok = false;
    RjGraphDev rj{};                      // the dynamics' view
    RjGraphDev rj_hi{}, rj_lo{};          // the same topology with the two integer levels of the ORIGINAL couplings (energies)
    int rj_k = 0;                         // the dynamics' couplings are integers in units of 2^rj_k (heavy sites: 2^(rj_k + dshift))
    int rj_k_energy = 0;                  // energy = 2^rj_k_energy S(hi) + 2^(rj_k_energy - 24) S(lo)
    uint32_t rj_heavy_sites = 0;
    bool stable_path = false;             // ISINGMC_FLAG_STABLE_PATH: the kernel family never depends on the number of experiments
    Options opt;                          // the environment's switches when the graph was created
    std::vector<uint32_t> class_real_end; // per colour class: end of its real sites (the padding follows)
    std::vector<void *> dev_allocs;

    ~isingmc_graph()
    {
        (void)hipSetDevice(device);
        (void)hipDeviceSynchronize(); // the blocks are recycled (cached_free): no kernel may still be reading the graph
        for (void *p : dev_allocs) (void)cached_free(p);
    }
};

struct isingmc_states {
    isingmc_graph *g = nullptr;
    Options opt; // the environment's switches when the container was created (isingmc_states_set_option changes one)
    size_t R = 0, cap = 0;
    uint32_t *d_state = nullptr;
    uint2 *d_keys = nullptr;
    uint64_t t = 0; // absolute timestep = Philox counter
    hipStream_t stream = nullptr;
    std::vector<hipStream_t> lanes; // sweep launches of disjoint replica blocks alternate over these (see run_steps)
    std::vector<hipEvent_t> lane_events;
    hipEvent_t fork_event = nullptr;
    size_t n_lanes = 1; // lanes in use by the current run_steps call
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool has_betas = false;
    std::vector<double> betas;
    LatThr *d_thr = nullptr;
    LatThrMC *d_thr_mc = nullptr; // per-replica thresholds of the multi-class kernels (has_betas on a field / open lattice)
    double *d_beta = nullptr;
    // measurement scratch
    unsigned long long *d_meas = nullptr; // lattice: [R][2]
    bool meas_zero = false;               // d_meas is known to be all zero (left so by the tempering measurement)
    double *d_pe = nullptr, *d_oe = nullptr;
    long long *d_pm = nullptr, *d_om = nullptr;
    uint32_t n_partials = 0;
    // replica-packed general path: one word per position = 32 replicas of a group
    bool packed = false;
    size_t groups = 0;
    size_t pk_bit0 = 0; // replica r of this shard is bit (r + pk_bit0) % 32 of group (r + pk_bit0) / 32 (shards cut GLOBAL groups)
    size_t pk_slots() const { return 32 * groups; } // counter slots: one per (group, bit), owned or not
    uint32_t *d_tab = nullptr; // threshold tables [groups or steps][PK_TAB_WORDS]
    bool rj = false;           // packed container on the real-coupling path (real_kernels.hpp) instead of the bit-sliced one
    RjBeta *d_rj_betas = nullptr; // per-replica acceptance scales [32 groups] (has_betas)
    unsigned long long *d_pk_slot_thr = nullptr; // on-stream tempering on the bit-sliced packed path: T_m per slot [32 groups][PK_MAX_DEG]
    size_t n_total = 0, first = 0; // this container is the shard [first, first + R) of n_total experiments
    // persistent strip kernel (strip_kernels.hpp): halo granules, error word, tag epoch
    unsigned long long *d_halo = nullptr;
    size_t halo_cap = 0; // granules allocated
    uint32_t *d_strip_err = nullptr;
    uint32_t strip_epoch = 0;
    unsigned long long *d_pt_mail = nullptr, *d_pt_round_counts = nullptr; // in-kernel exchange rounds (StripLadder)
    uint32_t *d_pt_perm2 = nullptr;
    unsigned long long *d_strip_fin = nullptr; // [cap] final-measurement counters of the strip kernel (zero between launches)
    bool strip_test_failed = false; // ISINGMC_STRIP_TEST_FAIL_ONCE has fired for this object
    bool strip_disabled = false;    // a strip launch of this object timed out once: the per-colour launches serve it from then on
    uint32_t *d_snapshot = nullptr; // the planes a synchronous call started from (restored when a strip launch gives up)
    size_t snapshot_cap = 0;
    bool meas_fresh = false; // the tempering send buffer holds the energies of the CURRENT configurations (written by the last strip launch)
    // sampling pipeline (isingmc_run_sampling): two slabs of samples in flight
    hipStream_t copy_stream = nullptr;
    hipEvent_t sample_ready[2] = {nullptr, nullptr}, sample_copied[2] = {nullptr, nullptr};
    uint32_t *d_samples[2] = {nullptr, nullptr}, *h_samples[2] = {nullptr, nullptr}; // h_*: pinned
    unsigned long long *d_sample_counts[2] = {nullptr, nullptr}, *h_counts[2] = {nullptr, nullptr};
    double *d_sample_e[2] = {nullptr, nullptr}, *h_e[2] = {nullptr, nullptr};
    long long *d_sample_m = nullptr;
    size_t sample_cap_words = 0, sample_cap_counts = 0, sample_cap_e = 0;
    // on-stream parallel tempering (isingmc_pt_*)
    bool pt_attached = false;
    PtDev pt{};
    double *d_pt_ladder = nullptr, *d_pt_local = nullptr, *d_pt_all = nullptr;
    uint64_t *d_pt_ladder_thr = nullptr;
    uint32_t *d_pt_perm = nullptr;
    unsigned long long *d_pt_counters = nullptr;
    size_t pt_per = 0, pt_world = 1;

    ~isingmc_states()
    {
        if (!g) return;
        (void)hipSetDevice(g->device);
        // the blocks below go back to the cache, where the next request may pick them up at once: nothing of this object may
        // still be running (hipFree used to wait for the whole device)
        if (stream) (void)stream_quiesce(stream);
        for (auto st : lanes) (void)stream_quiesce(st);
        if (copy_stream) (void)stream_quiesce(copy_stream);
        for (void *p : {(void *)d_state, (void *)d_keys, (void *)d_thr, (void *)d_beta, (void *)d_meas,
                        (void *)d_pe, (void *)d_oe, (void *)d_pm, (void *)d_om})
            if (p) (void)cached_free(p);
        if (d_tab) (void)cached_free(d_tab);
        if (d_rj_betas) (void)cached_free(d_rj_betas);
        if (d_pk_slot_thr) (void)cached_free(d_pk_slot_thr);
        if (d_thr_mc) (void)cached_free(d_thr_mc);
        for (int b = 0; b < 2; b++) {
            for (void *p : {(void *)d_samples[b], (void *)d_sample_counts[b], (void *)d_sample_e[b]})
                if (p) (void)cached_free(p);
            for (void *p : {(void *)h_samples[b], (void *)h_counts[b], (void *)h_e[b]})
                if (p) (void)cached_host_free(p);
            pooled_event_destroy(sample_ready[b], true);
            pooled_event_destroy(sample_copied[b], true);
        }
        if (d_sample_m) (void)cached_free(d_sample_m);
        pooled_stream_destroy(copy_stream);
        if (d_halo) (void)cached_free(d_halo);
        if (d_strip_err) (void)cached_free(d_strip_err);
        if (d_strip_fin) (void)cached_free(d_strip_fin);
        if (d_snapshot) (void)cached_free(d_snapshot);
        for (void *p : {(void *)d_pt_mail, (void *)d_pt_round_counts, (void *)d_pt_perm2})
            if (p) (void)cached_free(p);
        for (void *p : {(void *)d_pt_ladder, (void *)d_pt_local, (void *)d_pt_all, (void *)d_pt_ladder_thr, (void *)d_pt_perm,
                        (void *)d_pt_counters})
            if (p) (void)cached_free(p);
        for (auto st : lanes) pooled_stream_destroy(st);
        for (auto ev : lane_events) pooled_event_destroy(ev, true);
        pooled_event_destroy(fork_event, true);
        pooled_event_destroy(ev0, false);
        pooled_event_destroy(ev1, false);
        pooled_stream_destroy(stream);
    }
};

template <typename T>
static int dev_alloc(T **out, size_t count)
{
    *out = nullptr;
    HIP_TRY(cached_malloc(reinterpret_cast<void **>(out), std::max<size_t>(count, 1) * sizeof(T)));
    return ISINGMC_OK;
}

// device scratch of one API call: freed on every exit path, after the stream has drained
struct DeviceScratch {
    hipStream_t stream;
    std::vector<void *> ptrs;
    explicit DeviceScratch(hipStream_t st) : stream(st) {}
    DeviceScratch(const DeviceScratch &) = delete;
    ~DeviceScratch()
    {
        if (ptrs.empty()) return;
        (void)hipStreamSynchronize(stream);
        for (void *p : ptrs) (void)cached_free(p);
    }
    template <typename T>
    int alloc(T **out, size_t count)
    {
        TRY(dev_alloc(out, count));
        ptrs.push_back(*out);
        return ISINGMC_OK;
    }
};

template <typename T>
static int graph_upload(isingmc_graph *g, const T **dst, const std::vector<T> &src)
{
    T *d = nullptr;
    TRY(dev_alloc(&d, src.size()));
    g->dev_allocs.push_back(d);
    if (!src.empty()) HIP_TRY(hipMemcpy(d, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    *dst = d;
    return ISINGMC_OK;
}

template <typename F>
static void parallel_for(size_t n, F &&body, size_t bytes_per_item = size_t(1) << 20)
{
    // small jobs run on the calling thread: starting and joining threads costs ~100 us, more than expanding a few KB
    const size_t nthreads = n * bytes_per_item < (size_t(1) << 18)
                                ? 1 : std::min<size_t>(n, std::max(1u, std::min(32u, std::thread::hardware_concurrency())));
    if (nthreads <= 1) {
        for (size_t i = 0; i < n; i++) body(i);
        return;
    }
    std::vector<std::thread> pool;
    for (size_t tid = 0; tid < nthreads; tid++)
        pool.emplace_back([&, tid] {
            for (size_t i = tid; i < n; i += nthreads) body(i);
        });
    for (auto &th : pool) th.join();
}

// ---- graph.hip ---------------------------------------------------------------------------------------------------------------
IM_INTERNAL uint64_t threshold_fixed(double beta, double dE);
IM_INTERNAL LatThr lattice_thresholds(double beta, double jabs);
IM_INTERNAL LatThrMC lattice_thresholds_mc(const isingmc_graph *g, double beta);
IM_INTERNAL double lattice_energy(const isingmc_graph *g, unsigned long long sat, unsigned long long up);
IM_INTERNAL void pack_state(const isingmc_graph *g, const uint8_t *spins, uint32_t *words);
IM_INTERNAL void unpack_state(const isingmc_graph *g, const uint32_t *words, uint8_t *spins);

// ---- isingmc.hip (replica containers and launches) ---------------------------------------------------------------------------
constexpr size_t MAX_GRID_Y = 32768;
constexpr int STRIP_TIMED_OUT = 1000; // internal status, never returned through the C ABI
struct StripPlan {
    bool use = false;
    int nw = 1; // waves per strip (workgroup)
    StripArgs a{};
    size_t replicas_per_pass = 0; // a pass = one launch over a block of replicas for all timesteps of the chunk
};
IM_INTERNAL int run_steps(isingmc_states *s, size_t timesteps, const double *betas, size_t beta_stride, double *energies_per_step,
                          float *device_ms, bool sync = true, double *final_energies = nullptr);
IM_INTERNAL int set_betas(isingmc_states *s, const double *beta_per_replica, bool all_equal);
IM_INTERNAL int measure_enqueue(isingmc_states *s, unsigned long long *counts_slot, double *e_slot, long long *m_slot, bool want_up = true);
IM_INTERNAL void lat_measure_enqueue(isingmc_states *s, unsigned long long *out, size_t out_stride); // lattice containers: the counting launches alone
IM_INTERNAL double pk_energy(const isingmc_graph *g, bool rj, unsigned long long c0, unsigned long long c1);
IM_INTERNAL int pk_get_states(isingmc_states *s, uint8_t *states_out, size_t replica_stride_bytes, uint32_t *packed_out);
IM_INTERNAL StripPlan strip_plan(const isingmc_states *s, size_t timesteps, bool ladder = false);
IM_INTERNAL int launch_strip(isingmc_states *s, const StripPlan &P, size_t r0, size_t n, size_t nk, const LatThr *d_thr_steps, uint32_t thr_stride,
                             unsigned long long *steps_out, double *final_energies, const StripLadder *ladder = nullptr);
IM_INTERNAL int strip_check(isingmc_states *s);
IM_INTERNAL int strip_error(int rc);
IM_INTERNAL bool may_use_strips(const isingmc_states *s);
IM_INTERNAL int snapshot_take(isingmc_states *s);
IM_INTERNAL int snapshot_restore(isingmc_states *s);
