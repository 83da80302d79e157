// libisingmc.so: the C ABI of include/isingmc.h over the HIP kernels (gfx950 only).
// Host orchestration only -- every Monte-Carlo operation runs in the kernels of
// lattice_kernels.hpp / general_kernels.hpp.  There is no CPU fallback.
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

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

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

// ------------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Device blocks are recycled: a call of the reference's API creates its replicas, runs and drops them again
// (Lattice.run_monte_carlo, lattice.rs:171-221), and for a small lattice hipMalloc / hipFree -- each a device-wide
// synchronisation -- cost more than the timesteps: 2.1 ms of a 2.2 ms call of ONE timestep on 16 x 16 x 4 (tools/small_call_overhead.py).
// Freed blocks of up to 64 MiB wait in a per-device list (at most 512 MiB / 256 blocks) for the next request of exactly their
// size.  Every owner synchronises its streams before it frees (hipFree did that implicitly).  ISINGMC_NO_ALLOC_CACHE=1: off.
// ------------------------------------------------------------------------------------------------
namespace {
struct DevCache {
    std::mutex mu;
    std::unordered_map<void *, std::pair<int, size_t>> live;        // block -> (device, bytes)
    std::multimap<std::pair<int, size_t>, void *> idle;             // (device, bytes) -> block
    size_t idle_bytes = 0;
    static constexpr size_t MAX_BLOCK = size_t(64) << 20, MAX_IDLE = size_t(512) << 20, MAX_COUNT = 256;
};
DevCache &dev_cache()
{
    static DevCache *c = new DevCache; // never destroyed: the HIP runtime may be gone by the time static destructors run
    return *c;
}
bool dev_cache_off()
{
    static const bool off = [] { const char *e = std::getenv("ISINGMC_NO_ALLOC_CACHE"); return e && *e && *e != '0'; }();
    return off;
}
} // namespace

static hipError_t cached_malloc(void **out, size_t bytes)
{
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return err;
    DevCache &c = dev_cache();
    if (!dev_cache_off()) {
        std::lock_guard<std::mutex> lock(c.mu);
        auto it = c.idle.find({dev, bytes});
        if (it != c.idle.end()) {
            *out = it->second;
            c.idle.erase(it);
            c.idle_bytes -= bytes;
            c.live[*out] = {dev, bytes};
            return hipSuccess;
        }
    }
    err = hipMalloc(out, bytes);
    if (err != hipSuccess && !dev_cache_off()) { // out of memory: give the idle blocks back and try once more
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lock(c.mu);
            for (auto &kv : c.idle) drop.push_back(kv.second);
            c.idle.clear();
            c.idle_bytes = 0;
        }
        for (void *p : drop) (void)hipFree(p);
        (void)hipGetLastError();
        err = hipMalloc(out, bytes);
    }
    if (err == hipSuccess && !dev_cache_off()) {
        std::lock_guard<std::mutex> lock(c.mu);
        c.live[*out] = {dev, bytes};
    }
    return err;
}

static hipError_t cached_free(void *p)
{
    if (!p) return hipSuccess;
    DevCache &c = dev_cache();
    {
        std::lock_guard<std::mutex> lock(c.mu);
        auto it = c.live.find(p);
        if (it != c.live.end()) {
            const auto key = it->second;
            c.live.erase(it);
            if (!dev_cache_off() && key.second <= DevCache::MAX_BLOCK && c.idle_bytes + key.second <= DevCache::MAX_IDLE &&
                c.idle.size() < DevCache::MAX_COUNT) {
                c.idle.emplace(key, p);
                c.idle_bytes += key.second;
                return hipSuccess;
            }
        }
    }
    return hipFree(p);
}

// ... and pinned host blocks (the staging buffers of get_states / the sampling pipeline: pinning and unpinning cost ~150 us each)
namespace {
struct HostCache {
    std::mutex mu;
    std::unordered_map<void *, size_t> live;
    std::multimap<size_t, void *> idle;
    size_t idle_bytes = 0;
};
HostCache &host_cache()
{
    static HostCache *c = new HostCache;
    return *c;
}
} // namespace

static hipError_t cached_host_malloc(void **out, size_t bytes)
{
    HostCache &c = host_cache();
    if (!dev_cache_off()) {
        std::lock_guard<std::mutex> lock(c.mu);
        auto it = c.idle.find(bytes);
        if (it != c.idle.end()) {
            *out = it->second;
            c.idle.erase(it);
            c.idle_bytes -= bytes;
            c.live[*out] = bytes;
            return hipSuccess;
        }
    }
    const hipError_t err = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (err == hipSuccess && !dev_cache_off()) {
        std::lock_guard<std::mutex> lock(c.mu);
        c.live[*out] = bytes;
    }
    return err;
}

static hipError_t cached_host_free(void *p)
{
    if (!p) return hipSuccess;
    HostCache &c = host_cache();
    {
        std::lock_guard<std::mutex> lock(c.mu);
        auto it = c.live.find(p);
        if (it != c.live.end()) {
            const size_t bytes = it->second;
            c.live.erase(it);
            if (!dev_cache_off() && bytes <= (size_t(64) << 20) && c.idle_bytes + bytes <= (size_t(256) << 20) && c.idle.size() < 64) {
                c.idle.emplace(bytes, p);
                c.idle_bytes += bytes;
                return hipSuccess;
            }
        }
    }
    return hipHostFree(p);
}

// Streams are recycled the same way (creating and destroying the three streams of a replica container took ~1.5 ms of that
// call): non-blocking streams per device, handed back idle (their owner synchronises them first).
namespace {
struct StreamPool {
    std::mutex mu;
    std::multimap<int, hipStream_t> idle; // device -> stream
};
StreamPool &stream_pool()
{
    static StreamPool *p = new StreamPool;
    return *p;
}
} // namespace

static hipError_t pooled_stream_create(hipStream_t *out)
{
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return err;
    if (!dev_cache_off()) {
        StreamPool &p = stream_pool();
        std::lock_guard<std::mutex> lock(p.mu);
        auto it = p.idle.find(dev);
        if (it != p.idle.end()) {
            *out = it->second;
            p.idle.erase(it);
            return hipSuccess;
        }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

// hipStreamSynchronize costs ~70 us even on an idle stream; a query is enough when everything has completed
static hipError_t stream_quiesce(hipStream_t st)
{
    if (hipStreamQuery(st) == hipSuccess) return hipSuccess;
    (void)hipGetLastError(); // hipErrorNotReady is not an error
    return hipStreamSynchronize(st);
}

static void pooled_stream_destroy(hipStream_t st)
{
    if (!st) return;
    int dev = 0;
    if (!dev_cache_off() && stream_quiesce(st) == hipSuccess && hipGetDevice(&dev) == hipSuccess) {
        StreamPool &p = stream_pool();
        std::lock_guard<std::mutex> lock(p.mu);
        if (p.idle.size() < 64) {
            p.idle.emplace(dev, st);
            return;
        }
    }
    (void)hipStreamDestroy(st);
}

// ... and events (two kinds: with timing for the *_timed entry point, without for ordering between streams)
namespace {
struct EventPool {
    std::mutex mu;
    std::multimap<std::pair<int, bool>, hipEvent_t> idle; // (device, timing disabled?) -> event
};
EventPool &event_pool()
{
    static EventPool *p = new EventPool;
    return *p;
}
} // namespace

static hipError_t pooled_event_create(hipEvent_t *out, bool disable_timing)
{
    int dev = 0;
    if (!dev_cache_off() && hipGetDevice(&dev) == hipSuccess) {
        EventPool &p = event_pool();
        std::lock_guard<std::mutex> lock(p.mu);
        auto it = p.idle.find({dev, disable_timing});
        if (it != p.idle.end()) {
            *out = it->second;
            p.idle.erase(it);
            return hipSuccess;
        }
    }
    return disable_timing ? hipEventCreateWithFlags(out, hipEventDisableTiming) : hipEventCreate(out);
}

// (called with the owner's device current, as the destructors and creators here are)
static void pooled_event_destroy(hipEvent_t ev, bool disable_timing)
{
    if (!ev) return;
    int dev = 0;
    if (!dev_cache_off() && hipGetDevice(&dev) == hipSuccess) {
        EventPool &p = event_pool();
        std::lock_guard<std::mutex> lock(p.mu);
        if (p.idle.size() < 256) {
            p.idle.emplace(std::make_pair(dev, disable_timing), ev);
            return;
        }
    }
    (void)hipEventDestroy(ev);
}

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
    std::vector<uint32_t> pk_class_full; // per colour class: end of its last 256-block without padding
    uint64_t n_directed = 0;
    // replica-packed real-coupling path (real_kernels.hpp): any couplings and biases, degree <= 15
    bool rj_ok = false;
    RjGraphDev rj{};
    int rj_k = 0;                         // couplings are integers in units of 2^rj_k
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

static int use_device(int device)
{
    int count = 0;
    hipError_t err = hipGetDeviceCount(&count);
    if (err != hipSuccess || count <= 0)
        return fail(ISINGMC_ERR_NO_DEVICE,
                    std::string("no HIP device available (libisingmc has no CPU fallback): ") +
                        hipGetErrorString(err));
    if (device < 0 || device >= count)
        return fail(ISINGMC_ERR_NO_DEVICE, "device ordinal " + std::to_string(device) + " out of range (" +
                                               std::to_string(count) + " devices)");
    HIP_TRY(hipSetDevice(device));
    // the "last error" is per thread and shared with every other HIP user in the process (e.g. torch):
    // clear what others left behind so that the hipGetLastError() checks after our launches see only ours
    (void)hipGetLastError();
    return ISINGMC_OK;
}

// ------------------------------------------------------------------------------------------------
// small host helpers
// ------------------------------------------------------------------------------------------------
// acceptance probability as a THR_BITS-bit fixed-point threshold: accept iff u < T, u uniform on
// [0, 2^THR_BITS); T = 2^THR_BITS accepts always (dE <= 0, or beta < 0)
static uint64_t threshold_fixed(double beta, double dE)
{
    const uint64_t ONE = uint64_t(1) << THR_BITS;
    if (dE <= 0.0) return ONE;
    const double p = std::exp(-beta * dE);
    if (!(p < 1.0)) return ONE;
    return uint64_t(std::floor(std::ldexp(p, THR_BITS)));
}

static LatThr lattice_thresholds(double beta, double jabs)
{
    return LatThr{threshold_fixed(beta, 4.0 * jabs), threshold_fixed(beta, 8.0 * jabs)};
}

// thresholds of the multi-class kernels (classes: mc_types.hpp); same fixed-point rule, same exp as the two-class ones
static LatThrMC lattice_thresholds_mc(const isingmc_graph *g, double beta)
{
    LatThrMC t{};
    const int nc = g->mc_mode == MC_FIELD_OPEN ? 9 : g->mc_mode == MC_FIELD ? 6 : g->mc_mode == MC_ANISO ? 5 : 4;
    for (int c = 0; c < nc; c++) {
        double dE;
        if (g->mc_mode == MC_FIELD_OPEN) { // classes by m = sat - unsat and sigma = spin x sign of the site's field: |h| here
            const int m = c < 8 ? 1 + c / 2 : 0;
            const double sval = (c == 8 || (c & 1)) ? 1.0 : -1.0;
            dE = 2.0 * g->jabs * double(m) + 2.0 * std::fabs(g->field) * sval;
        } else if (g->mc_mode == MC_ANISO) {
            static const int mx[5] = {2, 2, 0, 2, -2}, my[5] = {2, 0, 2, -2, 2}; // (sat - unsat) per direction of the classes
            dE = 2.0 * g->jabs * double(mx[c]) + 2.0 * g->jabs_y * double(my[c]); // the oracle's expression, term by term
        } else if (g->mc_mode == MC_FIELD) {
            const int k = 2 + c / 2;
            const double sval = (c & 1) ? 1.0 : -1.0;
            dE = 2.0 * g->jabs * double(2 * k - 4) + 2.0 * g->field * sval; // the oracle's expression: 2|J|(sat - unsat) + 2 h s
        } else {
            dE = 2.0 * g->jabs * double(c + 1);
        }
        const uint64_t T = threshold_fixed(beta, dE);
        if (!(T >> THR_BITS)) t.costly |= 1u << c;
        t.hi[c] = uint32_t(T >> 32) & ((1u << N_PLANES) - 1);
        t.lo[c] = uint32_t(T);
    }
    return t;
}

// lattice energy from the integer counters: E = |J| (bonds - 2 satisfied) - h (2 up - N)   (exact in f64 for h = 0)
static double lattice_energy(const isingmc_graph *g, unsigned long long sat, unsigned long long up)
{
    if (g->mc_mode == MC_ANISO) { // sat = satisfied horizontal | satisfied vertical << 32; N bonds per direction
        const int64_t n = int64_t(g->nvars), sx = int64_t(sat & 0xFFFFFFFFull), sy = int64_t(sat >> 32);
        return g->jabs * double(n - 2 * sx) + g->jabs_y * double(n - 2 * sy);
    }
    if (g->d_fneg) { // sat = satisfied bonds | spins along their site's field << 32; field = |h|
        const int64_t k = int64_t(sat & 0xFFFFFFFFull), along = int64_t(sat >> 32);
        return g->jabs * double(int64_t(g->n_edges) - 2 * k) - g->field * double(2 * along - int64_t(g->nvars));
    }
    const double bonds = g->jabs * double(int64_t(g->n_edges) - 2 * int64_t(sat));
    if (g->mc_mode != MC_FIELD && g->mc_mode != MC_FIELD_OPEN) return bonds;
    return bonds - g->field * double(2 * int64_t(up) - int64_t(g->nvars));
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

// bytes (site order) -> packed words of one replica
static void pack_state(const isingmc_graph *g, const uint8_t *spins, uint32_t *words)
{
    std::fill(words, words + g->state_words, 0u);
    if (g->kind == ISINGMC_KIND_LATTICE2D) {
        const LatGeom &L = g->geom;
        for (uint32_t y = 0; y < L.H; y++)
            for (uint32_t x = 0; x < L.W; x++)
                if (spins[size_t(y) * L.W + x]) {
                    const uint32_t c = (x + y) & 1, i = x >> 1;
                    words[size_t(c) * L.wpp + size_t(y) * L.wpr + (i >> 5)] |= 1u << (i & 31);
                }
    } else {
        for (uint64_t i = 0; i < g->nvars; i++)
            if (spins[i]) words[g->pos[i] >> 5] |= 1u << (g->pos[i] & 31);
    }
}

// packed words of one replica -> bytes (site order)
static void unpack_state(const isingmc_graph *g, const uint32_t *words, uint8_t *spins)
{
    if (g->kind == ISINGMC_KIND_LATTICE2D) {
        unpack_lattice(g->geom.W, g->geom.H, words, spins);
    } else {
        for (uint64_t i = 0; i < g->nvars; i++) spins[i] = (words[g->pos[i] >> 5] >> (g->pos[i] & 31)) & 1u;
    }
}

// ------------------------------------------------------------------------------------------------
// C ABI: misc + host-only helpers
// ------------------------------------------------------------------------------------------------
extern "C" const char *isingmc_last_error(void) { return g_last_error.c_str(); }

extern "C" int isingmc_abi_version(void) { return ISINGMC_ABI_VERSION; }

extern "C" size_t isingmc_release_cached_resources(void)
{
    size_t bytes = 0;
    std::vector<void *> dev_blocks, host_blocks;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> events;
    {
        DevCache &c = dev_cache();
        std::lock_guard<std::mutex> lock(c.mu);
        for (auto &kv : c.idle) dev_blocks.push_back(kv.second);
        bytes += c.idle_bytes;
        c.idle.clear();
        c.idle_bytes = 0;
    }
    {
        HostCache &c = host_cache();
        std::lock_guard<std::mutex> lock(c.mu);
        for (auto &kv : c.idle) host_blocks.push_back(kv.second);
        bytes += c.idle_bytes;
        c.idle.clear();
        c.idle_bytes = 0;
    }
    {
        StreamPool &p = stream_pool();
        std::lock_guard<std::mutex> lock(p.mu);
        for (auto &kv : p.idle) streams.push_back(kv.second);
        p.idle.clear();
    }
    {
        EventPool &p = event_pool();
        std::lock_guard<std::mutex> lock(p.mu);
        for (auto &kv : p.idle) events.push_back(kv.second);
        p.idle.clear();
    }
    for (void *b : dev_blocks) (void)hipFree(b);
    for (void *b : host_blocks) (void)hipHostFree(b);
    for (hipStream_t st : streams) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
    (void)hipGetLastError();
    return bytes;
}

extern "C" int isingmc_device_count(int *count)
{
    if (!count) return fail(ISINGMC_ERR_INVALID, "count is NULL");
    *count = 0;
    hipError_t err = hipGetDeviceCount(count);
    if (err != hipSuccess) {
        *count = 0;
        return fail(ISINGMC_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(err));
    }
    return ISINGMC_OK;
}

extern "C" int isingmc_host_make_seeds(int has_seed, uint64_t seed_gen, size_t n, uint64_t *seeds_out)
{
    if (n && !seeds_out) return fail(ISINGMC_ERR_INVALID, "seeds_out is NULL");
    const auto seeds = make_seeds(has_seed != 0, seed_gen, n);
    std::copy(seeds.begin(), seeds.end(), seeds_out);
    return ISINGMC_OK;
}

extern "C" int isingmc_host_expand_schedule(const uint64_t *stop_t, const double *stop_beta, size_t n_stops,
                                            size_t timesteps, int compat_constant_beta, double *betas_out)
{
    if ((n_stops && (!stop_t || !stop_beta)) || (timesteps && !betas_out))
        return fail(ISINGMC_ERR_INVALID, "NULL schedule argument");
    const std::string msg = expand_schedule(stop_t, stop_beta, n_stops, timesteps, compat_constant_beta != 0, betas_out);
    return msg.empty() ? ISINGMC_OK : fail(ISINGMC_ERR_INVALID, msg);
}

static int check_edges(const uint64_t *ea, const uint64_t *eb, const double *ej, size_t n_edges, size_t nvars)
{
    if (n_edges == 0) return fail(ISINGMC_ERR_INVALID, "Must supply some edges for graph"); // lattice.rs:70-72
    if (!ea || !eb || !ej) return fail(ISINGMC_ERR_INVALID, "NULL edge array");
    if (nvars == 0 || nvars > 0xFFFFFFF0ull) return fail(ISINGMC_ERR_INVALID, "nvars out of range (1 .. 2^32-16)");
    for (size_t k = 0; k < n_edges; k++) {
        if (ea[k] >= nvars || eb[k] >= nvars)
            return fail(ISINGMC_ERR_INVALID, "Index out of bounds: edge " + std::to_string(k) + " touches variable " +
                                                 std::to_string(std::max(ea[k], eb[k])) + " out of " + std::to_string(nvars));
        if (!std::isfinite(ej[k])) return fail(ISINGMC_ERR_INVALID, "edge couplings must be finite");
    }
    return ISINGMC_OK;
}

extern "C" int isingmc_host_recognise_lattice2d(const uint64_t *ea, const uint64_t *eb, const double *ej,
                                                size_t n_edges, size_t nvars, int *is_lattice, int *width,
                                                int *height, double *jabs, int *uniform_sign)
{
    if (!is_lattice) return fail(ISINGMC_ERR_INVALID, "is_lattice is NULL");
    TRY(check_edges(ea, eb, ej, n_edges, nvars));
    const Lattice2D L = recognise_lattice2d(ea, eb, ej, n_edges, nvars);
    // bit 0: a W x H lattice; bits 1, 2: open in x, y; bit 3: |J| differs between the directions
    *is_lattice = L.ok ? 1 + 2 * int(L.open_x) + 4 * int(L.open_y) + 8 * int(L.jabs != L.jabs_y) : 0;
    if (width) *width = L.W;
    if (height) *height = L.H;
    if (jabs) *jabs = L.jabs;
    if (uniform_sign) *uniform_sign = L.uniform_sign;
    return ISINGMC_OK;
}

extern "C" int isingmc_host_colour_graph(const uint64_t *ea, const uint64_t *eb, size_t n_edges, size_t nvars,
                                         uint32_t *colours_out, uint32_t *n_colours_out)
{
    std::vector<double> ones(n_edges, 1.0);
    TRY(check_edges(ea, eb, ones.data(), n_edges, nvars));
    const Adjacency A = build_adjacency(ea, eb, ones.data(), n_edges, nvars);
    const Colouring C = greedy_colouring(A, nvars);
    if (colours_out) std::copy(C.colour.begin(), C.colour.end(), colours_out);
    if (n_colours_out) *n_colours_out = C.n_colours;
    return ISINGMC_OK;
}

extern "C" int isingmc_host_pt_swap_round(uint64_t seed, uint64_t round, size_t n_rungs, const double *betas,
                                          const double *slot_energy, uint32_t *perm, uint64_t *swaps_out)
{
    if (n_rungs && (!betas || !slot_energy || !perm)) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    for (size_t i = 0; i < n_rungs; i++)
        if (perm[i] >= n_rungs) return fail(ISINGMC_ERR_INVALID, "perm is not a permutation of the rungs");
    const uint64_t swaps = pt_swap_round(seed, round, n_rungs, betas, slot_energy, perm);
    if (swaps_out) *swaps_out = swaps;
    return ISINGMC_OK;
}

extern "C" int isingmc_host_rj_quantise(const uint64_t *ea, const uint64_t *eb, const double *ej, size_t n_edges, size_t nvars,
                                        const double *biases, int32_t *jq_out, int32_t *hq_out, int *k_out, int *eligible_out)
{
    TRY(check_edges(ea, eb, ej, n_edges, nvars));
    if (biases)
        for (size_t i = 0; i < nvars; i++)
            if (!std::isfinite(biases[i])) return fail(ISINGMC_ERR_INVALID, "biases must be finite");
    const Adjacency A = build_adjacency(ea, eb, ej, n_edges, nvars);
    const RjQuant Q = rj_quantise(A, nvars, biases);
    if (k_out) *k_out = Q.k;
    if (eligible_out) *eligible_out = Q.eligible;
    if (hq_out) std::copy(Q.hq.begin(), Q.hq.end(), hq_out);
    if (jq_out) { // adjacency order -> input-edge order: edge e is the next unfilled entry of its first end's row
        std::vector<uint64_t> fill(A.ptr.begin(), A.ptr.end());
        for (size_t e = 0; e < n_edges; e++) {
            if (ea[e] == eb[e]) { jq_out[e] = 0; continue; }
            jq_out[e] = Q.jq[fill[ea[e]]];
            fill[ea[e]]++;
            fill[eb[e]]++;
        }
    }
    return ISINGMC_OK;
}

extern "C" int isingmc_host_rj_beta(double beta, int k, uint32_t *shift_out, uint32_t *mant_out)
{
    if (!shift_out || !mant_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    if (!std::isfinite(beta)) return fail(ISINGMC_ERR_INVALID, "beta must be finite");
    rj_beta(beta, k, shift_out, mant_out);
    return ISINGMC_OK;
}

extern "C" int isingmc_host_rj_log_table(uint32_t *table_out)
{
    if (!table_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    rj_log_table(table_out);
    return ISINGMC_OK;
}

// ------------------------------------------------------------------------------------------------
// graph
// ------------------------------------------------------------------------------------------------
// the uniform field h (0 without); fields of one size and both signs (h_i = +-h): |h| and *signs = true;
// NaN when the biases differ from site to site in any other way
static double uniform_bias(const double *biases, size_t nvars, bool *signs)
{
    *signs = false;
    if (!biases) return 0.0;
    bool equal = true, same_size = true;
    for (size_t i = 1; i < nvars; i++) {
        equal &= biases[i] == biases[0];
        same_size &= std::fabs(biases[i]) == std::fabs(biases[0]);
    }
    if (equal) return biases[0];
    if (!same_size) return std::numeric_limits<double>::quiet_NaN();
    *signs = true;
    return std::fabs(biases[0]);
}

// Periodic and field-free: the two-class kernels of lattice_kernels.hpp.  A field |h| <= 2|J| (uniform, or +-h from
// site to site) on a periodic lattice, open boundaries without a field or with a field |h| <= |J|, anisotropic couplings
// (periodic, no field): the multi-class kernels (whole quads per row needed).  Anything else (other site-dependent
// biases, larger fields, anisotropy with a field or open boundaries): the general path.
static bool lattice_fast_path_ok(const Lattice2D &L, double h, bool field_signs)
{
    if (!L.ok || L.W % 64 != 0) return false;
    if (std::isnan(h)) return false;
    const bool open = L.open_x || L.open_y, aniso = L.jabs != L.jabs_y;
    if (aniso && (open || h != 0.0)) return false; // anisotropic couplings: periodic and field-free only
    if (field_signs && uint64_t(L.W) * uint64_t(L.H) >= (uint64_t(1) << 31)) return false; // two 32-bit counters in one word
    if (h != 0.0 && !(std::fabs(h) <= 2.0 * L.jabs)) return false;
    // a boundary site with one more unsatisfied than satisfied bond (m = -1) must still flip outright: |h| <= |J| there
    if (open && h != 0.0 && !(std::fabs(h) <= L.jabs)) return false;
    if ((open || h != 0.0 || aniso) && (L.W / 64) % 4 != 0) return false;
    if (aniso && uint64_t(L.W) * uint64_t(L.H) >= (uint64_t(1) << 32)) return false; // two 32-bit bond counters in one word
    const uint64_t wpp = uint64_t(L.H) * uint64_t(L.W / 64);
    // the kernels address a replica through ONE buffer descriptor (int num_records) and 32-bit byte offsets:
    // both planes must fit below 2^31 bytes; larger lattices take the general path
    return wpp % 4 == 0 && 2 * wpp * sizeof(uint32_t) < (uint64_t(1) << 31);
}

static int build_lattice(isingmc_graph *g, const Lattice2D &L, double h, const double *biases, bool field_signs)
{
    g->kind = ISINGMC_KIND_LATTICE2D;
    const bool open = L.open_x || L.open_y;
    g->mc_mode = h != 0.0 ? (open ? MC_FIELD_OPEN : MC_FIELD) : open ? MC_OPEN : L.jabs != L.jabs_y ? MC_ANISO : MC_NONE;
    g->jabs_y = L.jabs_y;
    g->field = h; // signed for a uniform field, |h| with sign planes
    g->open = McOpen{uint32_t(L.open_x), uint32_t(L.open_y), (!field_signs && h < 0.0) ? 0xFFFFFFFFu : 0u};
    LatGeom &G = g->geom;
    G.W = L.W;
    G.H = L.H;
    G.wpr = L.W / 64;
    G.wpp = G.H * G.wpr;
    G.nquads = G.wpp / 4;
    g->vec = G.wpr % 4 == 0;
    G.cols_log2 = -1;
    if (g->vec) { // division-free, parity-uniform thread mapping (thread_to_quad)
        const uint32_t cols = G.wpr / 4;
        if ((cols & (cols - 1)) == 0) {
            int cl = 0;
            while ((1u << cl) < cols) cl++;
            const uint32_t rows_per_pair = cl >= 6 ? 1 : 2 * (64u >> cl);
            if (G.H % rows_per_pair == 0 && G.nquads % 64 == 0) G.cols_log2 = cl;
        }
    }
    g->jabs = L.jabs;
    g->uniform_sign = L.uniform_sign;
    g->jneg_uniform = L.jpos_uniform ? 0u : 0xFFFFFFFFu;
    g->state_words = 2 * uint64_t(G.wpp);
    g->n_colours = 2;
    if (!L.uniform_sign) { // per-bond sign planes in each colour's compact layout
        std::vector<uint32_t> jneg(size_t(8) * G.wpp, 0u);
        const uint32_t W = G.W, H = G.H;
        for (uint32_t c = 0; c < 2; c++)
            for (uint32_t y = 0; y < H; y++) {
                const uint32_t yu = (y + H - 1) % H, o = (y + c) & 1;
                for (uint32_t i = 0; i < W / 2; i++) {
                    const uint32_t x = 2 * i + o, xl = (x + W - 1) % W;
                    const bool up = L.jdown[size_t(yu) * W + x], dn = L.jdown[size_t(y) * W + x];
                    const bool left = L.jright[size_t(y) * W + xl], right = L.jright[size_t(y) * W + x];
                    const bool ce = o ? left : right, si = o ? right : left;
                    const size_t w = size_t(y) * G.wpr + (i >> 5);
                    const uint32_t bit = 1u << (i & 31);
                    uint32_t *base = jneg.data() + size_t(c) * 4 * G.wpp;
                    if (!up) base[w] |= bit;
                    if (!dn) base[G.wpp + w] |= bit;
                    if (!ce) base[2 * size_t(G.wpp) + w] |= bit;
                    if (!si) base[3 * size_t(G.wpp) + w] |= bit;
                }
            }
        const uint32_t *d = nullptr;
        TRY(graph_upload(g, &d, jneg));
        g->d_jneg = const_cast<uint32_t *>(d);
    }
    if (field_signs) { // bit set where h_i < 0, each colour's compact layout
        std::vector<uint32_t> fneg(size_t(2) * G.wpp, 0u);
        for (uint32_t c = 0; c < 2; c++)
            for (uint32_t y = 0; y < G.H; y++) {
                const uint32_t o = (y + c) & 1;
                for (uint32_t i = 0; i < G.W / 2; i++)
                    if (biases[size_t(y) * G.W + 2 * i + o] < 0.0) fneg[size_t(c) * G.wpp + size_t(y) * G.wpr + (i >> 5)] |= 1u << (i & 31);
            }
        const uint32_t *d = nullptr;
        TRY(graph_upload(g, &d, fneg));
        g->d_fneg = const_cast<uint32_t *>(d);
    }
    return ISINGMC_OK;
}

static bool env_flag(const char *name);

static int build_general(isingmc_graph *g, const uint64_t *ea, const uint64_t *eb, const double *ej,
                         size_t n_edges, size_t nvars, const double *biases)
{
    g->kind = ISINGMC_KIND_GENERAL;
    const Adjacency A = build_adjacency(ea, eb, ej, n_edges, nvars);
    if (A.nbr.size() >= 0xFFFFFFFFull) return fail(ISINGMC_ERR_INVALID, "too many edges for the general path (2^32 directed)");
    const Colouring C = greedy_colouring(A, nvars);
    if (C.n_pos >= 0xFFFFFFC0ull) return fail(ISINGMC_ERR_INVALID, "too many sites for the general path");
    g->self_energy = A.self_energy;
    g->n_colours = C.n_colours;
    g->class_base = C.class_base;
    g->pos = C.pos;
    g->state_words = C.n_pos / 32;

    const uint32_t n_pos = uint32_t(C.n_pos);
    std::vector<uint32_t> site(n_pos, PAD_SITE), rowptr(size_t(n_pos) + 1, 0);
    for (size_t i = 0; i < nvars; i++) site[C.pos[i]] = uint32_t(i);
    for (uint32_t p = 0; p < n_pos; p++)
        rowptr[p + 1] = rowptr[p] + (site[p] == PAD_SITE ? 0u : uint32_t(A.ptr[site[p] + 1] - A.ptr[site[p]]));
    std::vector<uint32_t> nbr(A.nbr.size());
    std::vector<double> w(A.w.size());
    bool lossless = true;
    for (uint32_t p = 0; p < n_pos; p++) {
        if (site[p] == PAD_SITE) continue;
        uint32_t o = rowptr[p];
        for (uint64_t e = A.ptr[site[p]]; e < A.ptr[site[p] + 1]; e++, o++) {
            nbr[o] = uint32_t(C.pos[A.nbr[e]]);
            w[o] = A.w[e];
            lossless &= double(float(A.w[e])) == A.w[e];
        }
    }
    GenGraphDev &D = g->gdev;
    D.n_pos = n_pos;
    D.n_words = n_pos / 32;
    g->gen_edges2 = uint32_t(nbr.size());
    TRY(graph_upload(g, &D.rowptr, rowptr));
    TRY(graph_upload(g, &D.nbr, nbr));
    TRY(graph_upload(g, &D.site, site));
    {
        std::vector<uint32_t> cb(C.class_base.begin(), C.class_base.end());
        TRY(graph_upload(g, &D.class_base, cb));
        D.n_colours = C.n_colours;
    }
    g->w_is_float = lossless;
    if (lossless) { // stream 4-byte couplings when that loses nothing (e.g. J = +-1)
        std::vector<float> wf(w.begin(), w.end());
        const float *d = nullptr;
        TRY(graph_upload(g, &d, wf));
        D.w = d;
    } else {
        const double *d = nullptr;
        TRY(graph_upload(g, &d, w));
        D.w = d;
    }
    // replica-packed eligibility: one |J| for every bond, no fields, degree <= PK_MAX_DEG
    {
        uint64_t maxdeg = 0;
        for (size_t i = 0; i < nvars; i++) maxdeg = std::max(maxdeg, A.ptr[i + 1] - A.ptr[i]);
        bool uniform = !w.empty() && !g->has_bias && maxdeg <= PK_MAX_DEG && n_pos < 0x80000000u;
        const double jabs = w.empty() ? 0.0 : std::fabs(w[0]);
        for (double x : w) uniform &= std::fabs(x) == jabs;
        uniform &= jabs > 0.0;
        if (uniform) {
            std::vector<uint32_t> ell(size_t(PK_MAX_DEG) * n_pos, PK_NO_NBR); // slot-major: coalesced per slot
            for (uint32_t p = 0; p < n_pos; p++)
                for (uint32_t e = rowptr[p]; e < rowptr[p + 1]; e++)
                    ell[size_t(e - rowptr[p]) * n_pos + p] = nbr[e] | (w[e] > 0.0 ? 0x80000000u : 0u);
            // block headers: a slot whose 64 entries of a block are one translation (or all unused) needs no table read
            const size_t n_blocks = n_pos / 64;
            std::vector<uint2> hdr(n_blocks * PK_MAX_DEG);
            parallel_for(n_blocks, [&](size_t B) {
                for (uint32_t i = 0; i < uint32_t(PK_MAX_DEG); i++) {
                    const uint32_t *e = ell.data() + size_t(i) * n_pos + 64 * B;
                    const uint32_t p0 = uint32_t(64 * B);
                    bool unused = true, uniform = e[0] != PK_NO_NBR;
                    const uint32_t sign = e[0] & 0x80000000u, delta = (e[0] & 0x7FFFFFFFu) - p0;
                    for (uint32_t l = 0; l < 64; l++) {
                        unused &= e[l] == PK_NO_NBR;
                        uniform &= e[l] != PK_NO_NBR && (e[l] & 0x80000000u) == sign && (e[l] & 0x7FFFFFFFu) - (p0 + l) == delta;
                    }
                    hdr[B * PK_MAX_DEG + i] = unused ? make_uint2(PK_HDR_UNUSED, 0) : uniform ? make_uint2(PK_HDR_UNIFORM | sign, delta)
                                                                                           : make_uint2(PK_HDR_MIXED, 0);
                }
            });
            PkGraphDev &P = g->pk;
            TRY(graph_upload(g, &P.ell_hdr, hdr));
            TRY(graph_upload(g, &P.nbr_ell, ell));
            P.site = D.site;
            P.class_base = D.class_base;
            P.n_colours = D.n_colours;
            P.n_pos = n_pos;
            g->packed_ok = true;
            g->jabs = jabs;
            g->n_directed = nbr.size();
            // one degree, one sign?  (isolated sites have degree 0: they rule the uniform kernel out too)
            uint64_t mindeg = maxdeg;
            for (size_t i = 0; i < nvars; i++) mindeg = std::min(mindeg, A.ptr[i + 1] - A.ptr[i]);
            bool one_sign = true;
            for (double x : w) one_sign &= (x > 0.0) == (w[0] > 0.0);
            if (mindeg == maxdeg && maxdeg >= 3) {
                g->pk_uni_deg = int(maxdeg);
                g->pk_uni_pmj = !one_sign;
                g->pk_uni.negmask = w[0] > 0.0 ? 0u : 0xFFFFFFFFu;
                // this kernel's block headers: translations whatever the signs, the signs as one 64-bit mask per block and slot
                std::vector<uint2> shift(n_blocks * PK_MAX_DEG, make_uint2(PK_HDR_MIXED, 0)), sign(n_blocks * PK_MAX_DEG, make_uint2(0, 0));
                parallel_for(n_blocks, [&](size_t B) {
                    for (uint32_t i = 0; i < uint32_t(maxdeg); i++) {
                        const uint32_t *e = ell.data() + size_t(i) * n_pos + 64 * B;
                        const uint32_t p0 = uint32_t(64 * B), delta = (e[0] & 0x7FFFFFFFu) - p0;
                        bool translation = true;
                        uint64_t mask = 0;
                        for (uint32_t l = 0; l < 64; l++) {
                            translation &= e[l] != PK_NO_NBR && (e[l] & 0x7FFFFFFFu) - (p0 + l) == delta;
                            mask |= uint64_t(e[l] != PK_NO_NBR && (e[l] >> 31)) << l;
                        }
                        if (translation) shift[B * PK_MAX_DEG + i] = make_uint2(PK_HDR_UNIFORM, delta);
                        sign[B * PK_MAX_DEG + i] = make_uint2(uint32_t(mask), uint32_t(mask >> 32));
                    }
                });
                TRY(graph_upload(g, &g->pk_uni.shift, shift));
                TRY(graph_upload(g, &g->pk_uni.sign, sign));
                g->pk_class_full.resize(C.n_colours);
                for (uint32_t c = 0; c < C.n_colours; c++) { // real sites come first in a class, the padding after them
                    uint32_t real = 0;
                    while (uint32_t(C.class_base[c]) + real < uint32_t(C.class_base[c + 1]) && site[uint32_t(C.class_base[c]) + real] != PAD_SITE) real++;
                    g->pk_class_full[c] = uint32_t(C.class_base[c]) + real / 256 * 256;
                }
            }
        }
    }
    g->class_real_end.resize(C.n_colours);
    for (uint32_t c = 0; c < C.n_colours; c++) { // real sites come first in a class
        uint32_t real = 0;
        while (uint32_t(C.class_base[c]) + real < uint32_t(C.class_base[c + 1]) && site[uint32_t(C.class_base[c]) + real] != PAD_SITE) real++;
        g->class_real_end[c] = uint32_t(C.class_base[c]) + real;
    }
    // real-coupling packed path: whatever the bit-sliced packed path cannot take (couplings of several sizes, site
    // biases), degree <= 15, quantisation faithful (rj_quantise)
    // (ISINGMC_FORCE_REAL=1 at graph creation builds it for graphs the bit-sliced path takes, too: the same Hamiltonian through
    // the other acceptance rule, for cross-checks such as tools/highstat.py)
    if ((!g->packed_ok || env_flag("ISINGMC_FORCE_REAL")) && n_pos < 0x80000000u) {
        const RjQuant Q = rj_quantise(A, nvars, biases);
        if (Q.eligible) {
            const uint32_t slots = Q.max_degree <= 4 ? 4u : Q.max_degree <= 7 ? 7u : Q.max_degree <= 11 ? 11u : 15u;
            std::vector<uint32_t> enbr(size_t(slots) * n_pos);
            std::vector<int32_t> ejq(size_t(slots) * n_pos, 0), ehq(n_pos, 0);
            for (uint32_t i = 0; i < slots; i++)
                for (uint32_t p = 0; p < n_pos; p++) enbr[size_t(i) * n_pos + p] = p; // unused slots point at the own position
            for (uint32_t p = 0; p < n_pos; p++) {
                if (site[p] == PAD_SITE) continue;
                ehq[p] = Q.hq[site[p]];
                uint32_t i = 0;
                for (uint64_t e = A.ptr[site[p]]; e < A.ptr[site[p] + 1]; e++, i++) {
                    enbr[size_t(i) * n_pos + p] = uint32_t(C.pos[A.nbr[e]]);
                    ejq[size_t(i) * n_pos + p] = Q.jq[e];
                }
            }
            uint32_t lt[RJ_LOG_INTERVALS + 1];
            rj_log_table(lt);
            std::vector<uint2> logtab(RJ_LOG_INTERVALS);
            for (int i = 0; i < RJ_LOG_INTERVALS; i++) logtab[i] = make_uint2(lt[i], lt[i + 1] - lt[i]);
            RjGraphDev &J = g->rj;
            TRY(graph_upload(g, &J.nbr, enbr));
            TRY(graph_upload(g, &J.jq, ejq));
            TRY(graph_upload(g, &J.hq, ehq));
            TRY(graph_upload(g, &J.logtab, logtab));
            J.n_pos = n_pos;
            J.slots = slots;
            g->rj_k = Q.k;
            g->rj_ok = true;
            // the packed containers' common parts (random start, set_state, copy-out) read these
            g->pk.site = D.site;
            g->pk.class_base = D.class_base;
            g->pk.n_colours = D.n_colours;
            g->pk.n_pos = n_pos;
        }
    }
    D.bias = nullptr;
    if (g->has_bias) {
        std::vector<double> bias(n_pos, 0.0);
        for (size_t i = 0; i < nvars; i++) bias[C.pos[i]] = biases[i];
        TRY(graph_upload(g, &D.bias, bias));
    }
    return ISINGMC_OK;
}

extern "C" int isingmc_graph_create(const uint64_t *ea, const uint64_t *eb, const double *ej, size_t n_edges,
                                    size_t nvars, const double *biases, int device, unsigned flags,
                                    isingmc_graph **graph_out)
{
    if (!graph_out) return fail(ISINGMC_ERR_INVALID, "graph_out is NULL");
    *graph_out = nullptr;
    TRY(check_edges(ea, eb, ej, n_edges, nvars));
    bool has_bias = false;
    if (biases)
        for (size_t i = 0; i < nvars; i++) {
            if (!std::isfinite(biases[i])) return fail(ISINGMC_ERR_INVALID, "biases must be finite");
            has_bias |= biases[i] != 0.0;
        }
    TRY(use_device(device));
    auto g = std::make_unique<isingmc_graph>();
    g->device = device;
    g->nvars = nvars;
    g->n_edges = n_edges;
    g->has_bias = has_bias;
    Lattice2D L;
    bool field_signs = false;
    const double h = has_bias ? uniform_bias(biases, nvars, &field_signs) : 0.0;
    if (!(flags & ISINGMC_FLAG_FORCE_GENERAL) && !std::isnan(h)) L = recognise_lattice2d(ea, eb, ej, n_edges, nvars);
    if (lattice_fast_path_ok(L, h, field_signs)) TRY(build_lattice(g.get(), L, h, biases, field_signs));
    else TRY(build_general(g.get(), ea, eb, ej, n_edges, nvars, biases));
    *graph_out = g.release();
    return ISINGMC_OK;
}

extern "C" int isingmc_graph_info(const isingmc_graph *g, isingmc_graph_info_t *info)
{
    if (!g || !info) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    std::memset(info, 0, sizeof *info);
    info->kind = g->kind;
    info->device = g->device;
    info->nvars = g->nvars;
    info->n_edges = g->n_edges;
    if (g->kind == ISINGMC_KIND_LATTICE2D) {
        info->width = int32_t(g->geom.W);
        info->height = int32_t(g->geom.H);
        info->jabs = g->jabs;
        info->jabs_y = g->mc_mode == MC_ANISO ? g->jabs_y : g->jabs;
        info->uniform_sign = g->uniform_sign;
        info->fast_path = g->mc_mode;
        info->field = g->field;
        info->open_x = int32_t(g->open.open_x);
        info->open_y = int32_t(g->open.open_y);
        info->field_signs = g->d_fneg ? 1 : 0;
    }
    info->n_colours = g->n_colours;
    info->packed_degree = g->packed_ok ? g->pk_uni_deg : 0;
    info->real_slots = g->rj_ok ? int32_t(g->rj.slots) : 0;
    info->real_quantum_log2 = g->rj_ok ? g->rj_k : 0;
    info->state_words = g->state_words;
    return ISINGMC_OK;
}

extern "C" void isingmc_graph_destroy(isingmc_graph *g) { delete g; }

// ------------------------------------------------------------------------------------------------
// states
// ------------------------------------------------------------------------------------------------
static dim3 lat_grid(const isingmc_graph *g, uint32_t quads, size_t replicas)
{
    (void)g;
    return dim3((quads + 255) / 256, unsigned(replicas), 1);
}

constexpr size_t MAX_GRID_Y = 32768;

static int lanes_reserve(isingmc_states *s, size_t n);
static int lanes_fork(isingmc_states *s, size_t n);
static int lanes_join(isingmc_states *s);

// replica-packed general path (defined further down)
static int choose_packed(const isingmc_graph *g, size_t n_replicas);
static int pk_create(isingmc_states *s, const uint64_t *all_seeds, size_t first, size_t n, const uint8_t *initial_state);
static int pk_set_state(isingmc_states *s, size_t replica, const uint8_t *spins);
static int pk_set_betas(isingmc_states *s);
static int pk_append(isingmc_states *s, uint64_t seed, const uint8_t *initial_state);
static bool resident_disabled();

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
    if (const int mode = count ? choose_packed(g, n_total) : 0) {
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

static int set_betas(isingmc_states *s, const double *beta_per_replica, bool all_equal);

extern "C" int isingmc_states_set_betas(isingmc_states *s, const double *beta_per_replica)
{
    return set_betas(s, beta_per_replica, false);
}

// all_equal: the caller passes one beta R times (run_sampling) -- then a shard that cuts a replica group is fine
static int set_betas(isingmc_states *s, const double *beta_per_replica, bool all_equal)
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
static bool env_flag(const char *name)
{
    const char *e = std::getenv(name);
    return e && e[0] && e[0] != '0';
}

static int env_int(const char *name, int dflt)
{
    const char *e = std::getenv(name);
    return e && e[0] ? std::atoi(e) : dflt;
}

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
    const bool csr_resident = gen_resident_fits(g, n_replicas) && !resident_disabled();
    if (real_path) {
        if (!csr_resident) return n_replicas >= 2; // partial groups draw only their own replicas' Philox calls: 1.4x the CSR launches at 2, 2.0x at 4, 2.9x at 8
        return n_replicas >= 16 && (g->nvars >= 8000 || (g->nvars >= 1500 && work >= (uint64_t(3) << 19))); // 1.5 x 2^20: re-measured with the graph staged in LDS
    }
    return work >= (uint64_t(1) << (csr_resident ? 22 : 19));
}

// 0: one replica per word set (CSR kernels); 1: replica-packed bit-sliced path (S6); 2: replica-packed real-coupling path (S7)
static int choose_packed(const isingmc_graph *g, size_t n_replicas)
{
    if (g->rj_ok && !env_flag("ISINGMC_DISABLE_REAL") && (!g->packed_ok || env_flag("ISINGMC_FORCE_REAL"))) {
        if (env_flag("ISINGMC_FORCE_REAL")) return n_replicas > 0 ? 2 : 0;
        return packed_worth_it(g, n_replicas, true) ? 2 : 0;
    }
    if (!g->packed_ok || env_flag("ISINGMC_DISABLE_PACKED")) return 0;
    // pk_sweep_kernel addresses the ELL table through one buffer descriptor with 32-bit byte offsets
    if (uint64_t(g->pk.n_pos) * PK_MAX_DEG * sizeof(uint32_t) >= (uint64_t(1) << 31)) return 0;
    if (env_flag("ISINGMC_FORCE_PACKED")) return n_replicas > 0 ? 1 : 0;
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

// ClassicIsing.add_graph (classicising.rs:62-79) on a replica-packed container.  A group simulates all of its 32 bit
// positions from the moment it is created (the spec numbers ties / draws words over whole groups), so a replica appended
// into a partly filled group takes over the chain its bit position has been running since then -- a configuration
// evolved under the same dynamics, i.e. as good a start as a fresh random one, and what the oracle engines give for the
// final replica count; with an initial_state it is set explicitly.  Replica 32 g opens a new group keyed by its seed,
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
        if (s->d_rj_betas) { (void)cached_free(s->d_rj_betas); s->d_rj_betas = nullptr; }
        hipLaunchKernelGGL(pk_init_kernel, dim3((g->pk.n_pos + 255) / 256, 1), dim3(256), 0, s->stream, s->d_state, g->pk, s->d_keys,
                           uint32_t(s->groups));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s->stream));
        s->groups = groups;
    }
    s->R = s->cap = s->n_total = slot + 1;
    if (initial_state) TRY(pk_set_state(s, slot, initial_state));
    else if (s->rj && slot % 32 != 0) {
        // real-coupling path: the bits of a group this container does not own are not simulated (rj_sweep_kernel PARTIAL), so the
        // new replica's column holds whatever it held: it starts from its random start, like a replica appended on any other
        // path.  (Bit-sliced path: the unused replicas of a group ARE simulated -- the group's tie numbering needs them -- and
        // the new replica continues that trajectory.)
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
        HIP_TRY(hipStreamSynchronize(s->stream));
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
    static const int target_wgs = std::max(256, env_int("ISINGMC_REAL_TARGET_WGS", 3072));
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

static void pk_launch_timestep(isingmc_states *s, const uint32_t *tabs, uint32_t tab_stride, size_t gb, size_t ge, hipStream_t stream)
{
    const isingmc_graph *g = s->g;
    const bool no_uni = env_flag("ISINGMC_DISABLE_PACKED_UNIFORM"); // A/B switch: results are the same either way
    for (uint32_t c = 0; c < g->n_colours; c++) {
        const uint32_t b = uint32_t(g->class_base[c]), e = uint32_t(g->class_base[c + 1]);
        if (e == b) continue;
        // blocks of real sites of a one-degree graph: the specialised kernel; the class's padded tail (and
        // every other graph): the general one.  tab_stride == 0 <=> one table, one beta for every replica.
        const uint32_t mid = g->pk_uni_deg && !no_uni ? g->pk_class_full[c] : b;
        for (size_t g0 = gb; g0 < ge; g0 += MAX_GRID_Y) {
            const size_t ng = std::min(MAX_GRID_Y, ge - g0);
            if (mid > b)
                (void)pk_uni_launch_sweep(g->pk_uni_deg, tab_stride == 0, g->pk_uni_pmj, dim3((mid - b) / 1024 + ((mid - b) % 1024 != 0), unsigned(ng)),
                                          stream, s->d_state + g0 * g->pk.n_pos, g->pk, g->pk_uni, b, mid, s->t, s->d_keys + g0,
                                          tabs + (tab_stride ? g0 * tab_stride : 0), tab_stride);
            if (e > mid)
                hipLaunchKernelGGL(pk_sweep_kernel, dim3((e - mid) / 1024 + ((e - mid) % 1024 != 0), unsigned(ng)), dim3(256), 0, stream,
                                   s->d_state + g0 * g->pk.n_pos, g->pk, mid, e, s->t, s->d_keys + g0,
                                   tabs + (tab_stride ? g0 * tab_stride : 0), tab_stride);
        }
    }
}

static int measure_enqueue(isingmc_states *s, unsigned long long *counts_slot, double *e_slot, long long *m_slot, bool want_up = true);

// energy of one replica of a packed container from the first counter of its slot
static double pk_energy(const isingmc_graph *g, bool rj, unsigned long long c0)
{
    // real-coupling path: c0 = -2 x the energy in units of 2^k (an exact, even integer; rj_measure_kernel)
    if (rj) return std::ldexp(double(-(int64_t(c0) / 2)), g->rj_k) + g->self_energy;
    // bit-sliced path: E = |J| (undirected bonds - 2 satisfied) + self loops; c0 = directed satisfied count (doubled)
    return g->jabs * (double(int64_t(g->n_directed / 2)) - double(int64_t(c0))) + g->self_energy;
}

static int pk_measure(isingmc_states *s, double *energies, int64_t *mags)
{
    const isingmc_graph *g = s->g;
    const size_t R = s->R;
    TRY(measure_enqueue(s, s->d_meas, nullptr, nullptr, /*want_up=*/mags != nullptr));
    s->meas_zero = false;
    std::vector<unsigned long long> h(2 * s->pk_slots());
    HIP_TRY(hipMemcpyAsync(h.data(), s->d_meas, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (size_t r = 0; r < R; r++) {
        const size_t sl = r + s->pk_bit0;
        if (energies) energies[r] = pk_energy(g, s->rj, h[2 * sl]);
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
        const int forced = env_int("ISINGMC_PK_STREAMS", 0);
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
    if (want_lanes > 1) TRY(lanes_fork(s, want_lanes));
    const size_t n_lanes = s->n_lanes, per_lane = (s->groups + n_lanes - 1) / n_lanes;
    const auto launch_step = [&](size_t k) {
        for (size_t lane = 0; lane < n_lanes; lane++) {
            const size_t gb = lane * per_lane, ge = std::min(s->groups, gb + per_lane);
            if (gb >= ge) continue;
            hipStream_t st = n_lanes > 1 ? s->lanes[lane] : s->stream;
            if (s->rj) {
                if (s->has_betas) rj_launch_timestep(s, s->d_rj_betas, 32, gb, ge, st);
                else rj_launch_timestep(s, d_rj_steps + (beta_stride ? k : 0), 0, gb, ge, st);
            } else if (s->has_betas) pk_launch_timestep(s, s->d_tab, PK_TAB_WORDS, gb, ge, st);
            else pk_launch_timestep(s, d_step_tabs + (beta_stride ? k * PK_TAB_WORDS : 0), 0, gb, ge, st);
        }
    };
    for (size_t k0 = 0; k0 < timesteps && rc == ISINGMC_OK; k0 += chunk) {
        const size_t nk = std::min(chunk, timesteps - k0);
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
                    energies_per_step[r * timesteps + k0 + k] = pk_energy(g, s->rj, h_step_counts[(k * CS + r + s->pk_bit0) * 2]);
        } else if (k0 + nk < timesteps && !s->has_betas && beta_stride) {
            // the next chunk overwrites the step tables: every lane must have finished reading them
            if (s->n_lanes > 1) { TRY(lanes_join(s)); HIP_TRY(hipStreamSynchronize(s->stream)); TRY(lanes_fork(s, want_lanes)); }
            else HIP_TRY(hipStreamSynchronize(s->stream));
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
static int pk_get_states(isingmc_states *s, uint8_t *states_out, size_t replica_stride_bytes, uint32_t *packed_out)
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
            static const unsigned dbg_lds = [] { const char *e = getenv("ISINGMC_DEBUG_SWEEP_LDS"); return e ? unsigned(atoi(e)) : 0u; }();
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
                static const int forced = [] { const char *e = getenv("ISINGMC_SWEEP_ITERS"); return e ? atoi(e) : 0; }();
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

// ISINGMC_DISABLE_RESIDENT=1: always use the per-colour launches (A/B runs, parity tests of both paths)
static bool resident_disabled()
{
    const char *e = std::getenv("ISINGMC_DISABLE_RESIDENT");
    return e && e[0] && e[0] != '0';
}

// ------------------------------------------------------------------------------------------------
// persistent strip kernel (strip_kernels.hpp): when and how
// ------------------------------------------------------------------------------------------------
struct StripPlan {
    bool use = false;
    int nw = 1; // waves per strip (workgroup)
    StripArgs a{};
    size_t replicas_per_pass = 0; // a pass = one launch over a block of replicas for all timesteps of the chunk
};

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

static StripPlan strip_plan(const isingmc_states *s, size_t timesteps, bool ladder = false)
{
    StripPlan P;
    const isingmc_graph *g = s->g;
    const int mode = env_int("ISINGMC_STRIP", -1);
    if (mode == 0 || s->strip_disabled || g->kind != ISINGMC_KIND_LATTICE2D || !g->vec || timesteps < 2) return P;
    const uint32_t qpr = g->geom.wpr / 4;
    if ((qpr & (qpr - 1)) != 0 || qpr > 32) return P; // power of two, at least two rows per wave
    P.nw = env_int("ISINGMC_STRIP_NW", 4) == 1 ? 1 : 4; // measured on 1024^2 x 64: 9.6 us per timestep either way; with exchange rounds 11.5 (4) / 12.0 (1)
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
    if (mode != 1 && total > size_t(env_int("ISINGMC_STRIP_MAX_WG", int(limit)))) return P;
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
static int launch_strip(isingmc_states *s, const StripPlan &P, size_t r0, size_t n, size_t nk, const LatThr *d_thr_steps,
                        uint32_t thr_stride, unsigned long long *steps_out, double *final_energies, const StripLadder *ladder = nullptr)
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
    if (!s->strip_test_failed && env_flag("ISINGMC_STRIP_TEST_FAIL_ONCE")) {
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
constexpr int STRIP_TIMED_OUT = 1000; // internal status, never returned through the C ABI
static int strip_check(isingmc_states *s)
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

static int strip_error(int rc)
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
static int snapshot_take(isingmc_states *s)
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

static int snapshot_restore(isingmc_states *s)
{
    HIP_TRY(hipMemcpyAsync(s->d_state, s->d_snapshot, s->R * s->g->state_words * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
    return ISINGMC_OK;
}

static bool may_use_strips(const isingmc_states *s)
{
    return s && s->R && !s->packed && !s->strip_disabled && s->g->kind == ISINGMC_KIND_LATTICE2D && s->g->mc_mode == MC_NONE &&
           env_int("ISINGMC_STRIP", -1) != 0;
}

static int run_steps(isingmc_states *s, size_t timesteps, const double *betas, size_t beta_stride,
                     double *energies_per_step, float *device_ms, bool sync = true, double *final_energies = nullptr)
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
                          !resident_disabled();
    // per-step counters: 16 B per (step, replica) and counter slot, at most 32 MiB per chunk on each side of the bus
    const StripPlan strip = (lattice && !resident && !mc) ? strip_plan(s, timesteps) : StripPlan{};
    s->meas_fresh = false;
    const size_t step_slots = (energies_per_step && lattice && !resident && !strip.use && !mc) ? MEASURE_SLOTS : 1;
    size_t chunk = energies_per_step ? std::max<size_t>(1, std::min<size_t>(timesteps, (size_t(32) << 20) / (16 * R * step_slots))) : timesteps;
    const bool gen_resident = !lattice && gen_resident_fits(g, R) && !resident_disabled();
    // the multi-class modes' LDS-resident kernel: same size bound
    const bool mc_resident = mc && g->state_words * sizeof(uint32_t) <= LDS_RESIDENT_MAX_BYTES && g->geom.nquads <= 1024 && !resident_disabled();
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
        const char *e = std::getenv("ISINGMC_STREAMS");
        if (e) want_lanes = std::max(1, std::atoi(e));
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
            static const int spread_mode = env_int("ISINGMC_RESIDENT_SPREAD", 1);
            // eight lanes per quad only: with four or two (256 / 512 quads per colour, 1024 threads) the barriers of a 16-wave workgroup
            // cost more than the shorter chain saves (256^2 x 64: 5.7 against 5.1 us; ISINGMC_RESIDENT_LPQ=4 / 2 for A/B runs)
            static const int lpq_forced = env_int("ISINGMC_RESIDENT_LPQ", 0);
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
            static const int mc_spread_mode = env_int("ISINGMC_RESIDENT_SPREAD", 1);
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
            static const int stage_mode = env_int("ISINGMC_GEN_STAGE", -1);
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
static int measure_enqueue(isingmc_states *s, unsigned long long *counts_slot, double *e_slot, long long *m_slot, bool want_up)
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
            static int per_cu[4][2][2] = {};
            int pc;
            {
                std::lock_guard<std::mutex> lock(per_cu_mutex);
                int &slot = per_cu[g->rj.slots == 4 ? 0 : g->rj.slots == 7 ? 1 : g->rj.slots == 11 ? 2 : 3][bip][want_up];
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
                HIP_TRY(rj_launch_measure(dim3(unsigned(std::max<size_t>(gx, 1)), unsigned(ng)), s->stream, s->d_state + g0 * g->pk.n_pos, g->rj,
                                          g->pk.site, class0_end, scan_end, want_up, counts_slot + 2 * 32 * g0));
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
    const size_t slab_bytes = size_t(std::max(1, env_int("ISINGMC_SAMPLE_SLAB_BYTES", 64 << 20))); // (tests shrink it)
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
                energy = pk_energy(g, s->rj, h_counts[(k * CS + sl) * 2]);
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

extern "C" int isingmc_pt_attach(isingmc_states *s, const double *ladder_betas, size_t n_rungs, size_t slot_offset,
                                 size_t slots_per_rank, size_t world_size, uint64_t seed)
{
    if (!s || !ladder_betas) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    if (s->pt_attached) return fail(ISINGMC_ERR_INVALID, "a ladder is already attached");
    const bool rj_ladder = s->packed && s->rj, pk_ladder = s->packed && !s->rj;
    if (!s->packed && (s->g->kind != ISINGMC_KIND_LATTICE2D || s->g->mc_mode != MC_NONE))
        return fail(ISINGMC_ERR_INVALID, "on-stream tempering is implemented for periodic, field-free lattices and for the replica-packed "
                                         "paths (use the host swap step)");
    // the replicas of a bit-sliced group number their ties together: a shard must hold whole groups (distributed.block_size aligns them)
    if (pk_ladder && (s->pk_bit0 != 0 || ((s->first + s->R) % 32 != 0 && s->first + s->R != s->n_total)))
        return fail(ISINGMC_ERR_INVALID, "a tempering shard on the replica-packed path must start and end on multiples of 32 slots");
    if (slot_offset + s->R > n_rungs || s->R > slots_per_rank || slots_per_rank * world_size < n_rungs || n_rungs >= 0xFFFFFFFFull)
        return fail(ISINGMC_ERR_INVALID, "ladder / shard geometry mismatch");
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
                           s->R == s->pt.n_rungs && env_int("ISINGMC_PT_IN_KERNEL", 1) != 0;
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
        LAT_DISPATCH(launch_lat_measure, s, s->d_meas, size_t(2));
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
        HIP_TRY(rj_launch_energy_from_counts(s->stream, s->d_meas, uint32_t(s->pk_bit0), uint32_t(R), g->rj_k, g->self_energy,
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
// measurement hook: the shader clock the chip holds WHILE the sweep kernels run (bench.py reports it next
// to the vector-ALU bound).  One wave on a side stream stamps s_memtime (shader cycles) and s_memrealtime
// (100 MHz, MI355X_MICROARCH.md "DVFS give-back" item 6) around a sleep loop of `probe_ms`, while `timesteps`
// sweeps run on the engine's stream.  The probe's exit condition is the constant-rate counter: every wave leaves.
// ------------------------------------------------------------------------------------------------
__global__ void clock_probe_kernel(unsigned long long *out, const unsigned long long realtime_ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1;
    do {
        __builtin_amdgcn_s_sleep(100);
        r1 = __builtin_amdgcn_s_memrealtime();
    } while (r1 - r0 < realtime_ticks);
    out[0] = __builtin_amdgcn_s_memtime() - c0;
    out[1] = r1 - r0;
}

extern "C" int isingmc_debug_shader_clock(isingmc_states *s, size_t timesteps, double beta, double probe_ms, double *ghz_out)
{
    if (!s || !ghz_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    if (!(probe_ms > 0.0) || probe_ms > 100.0) return fail(ISINGMC_ERR_INVALID, "probe_ms must be in (0, 100]");
    TRY(use_device(s->g->device));
    DeviceScratch scratch(s->stream);
    unsigned long long *d_out = nullptr, h_out[2] = {0, 0};
    TRY(scratch.alloc(&d_out, 2));
    struct Side { // a stream of its own: the replica lanes carry sweeps
        hipStream_t st = nullptr;
        ~Side() { if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } }
    } side_owner;
    HIP_TRY(hipStreamCreateWithFlags(&side_owner.st, hipStreamNonBlocking));
    hipStream_t side = side_owner.st;
    // the sweeps first (they ramp the chip up), then the probe beside them; both are awaited
    int rc = run_steps(s, timesteps / 4, &beta, 0, nullptr, nullptr, /*sync=*/false);
    if (rc != ISINGMC_OK) return rc;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, side, d_out, (unsigned long long)(probe_ms * 1e5));
    HIP_TRY(hipGetLastError());
    rc = run_steps(s, timesteps - timesteps / 4, &beta, 0, nullptr, nullptr, /*sync=*/true);
    HIP_TRY(hipStreamSynchronize(side));
    if (rc != ISINGMC_OK) return rc;
    HIP_TRY(hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost));
    *ghz_out = h_out[1] ? double(h_out[0]) / double(h_out[1]) * 0.1 : 0.0; // cycles per 10 ns tick -> GHz
    return ISINGMC_OK;
}
