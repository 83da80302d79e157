// libisingmc.so: error state, the caches of device blocks / pinned blocks / streams / events, environment helpers, and the C ABI
// entry points that need nothing else (see internal.hpp for the layout of the host translation units).
#include "internal.hpp"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

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

hipError_t cached_malloc(void **out, size_t bytes)
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

hipError_t cached_free(void *p)
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

hipError_t cached_host_malloc(void **out, size_t bytes)
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

hipError_t cached_host_free(void *p)
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

hipError_t pooled_stream_create(hipStream_t *out)
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
hipError_t stream_quiesce(hipStream_t st)
{
    if (hipStreamQuery(st) == hipSuccess) return hipSuccess;
    (void)hipGetLastError(); // hipErrorNotReady is not an error
    return hipStreamSynchronize(st);
}

void pooled_stream_destroy(hipStream_t st)
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

hipError_t pooled_event_create(hipEvent_t *out, bool disable_timing)
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
void pooled_event_destroy(hipEvent_t ev, bool disable_timing)
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

int use_device(int device)
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

bool env_flag(const char *name)
{
    const char *e = std::getenv(name);
    return e && e[0] && e[0] != '0';
}

int env_int(const char *name, int dflt)
{
    const char *e = std::getenv(name);
    return e && e[0] ? std::atoi(e) : dflt;
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

