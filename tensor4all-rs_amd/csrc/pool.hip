// pool.hip — process-wide caches of the HIP resources a handle needs: device blocks, pinned host blocks, streams, device facts.
//
// A solve of a small problem (BASELINE configs[1]) is a few milliseconds; creating its handle (two streams, a device-property
// query, some sixty hipMalloc / hipHostMalloc calls as the buffers grow) and destroying it again cost 1.9 + 4.1 ms around a
// 4.2 ms solve, and an adaptive patch farm creates a handle per patch.  Blocks are cached in power-of-two size classes and
// never handed back to the driver (bounded: very large blocks bypass the cache, and the cache stops growing at 16 GiB);
// streams are recycled after a synchronisation.  hipFree's implicit device synchronisation — which the callers' "free, then
// allocate a larger buffer" pattern relied on — is kept explicitly.  T4A_NO_POOL=1 restores plain hipMalloc / hipFree.
#include "common.hpp"

#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace t4a {
namespace pool {

namespace {

std::mutex g_mu;
std::mutex g_capture_mu; // a device-wide synchronisation must not run while any of our streams records a graph: it would invalidate the capture
struct PerDevice {
    std::map<size_t, std::vector<void*>> dev_free;  // class bytes -> blocks
    std::vector<hipStream_t> streams[3];            // 0 default priority, 1 highest, 2 lowest
    int cus = 0;
};
std::map<int, PerDevice> g_dev;
std::map<size_t, std::vector<void*>> g_pin_free;  // pinned host memory is not per device
size_t g_cached_bytes = 0;      // device blocks in the free lists
size_t g_cached_pin_bytes = 0;  // pinned blocks in the free list (page-locked memory is the scarcer resource: its own, lower cap)
constexpr size_t kBypass = (size_t)256 << 20;  // blocks of this size and more go straight to the driver
constexpr size_t kCacheLimit = (size_t)16 << 30;
constexpr size_t kPinCacheLimit = (size_t)1 << 30;
// this thread is recording a graph (it holds g_capture_mu): a release must neither wait for the device (it would invalidate the
// capture) nor take g_capture_mu again (self-deadlock).  Blocks released meanwhile are parked and handed in at capture_end().
thread_local int t_idle_scope = 0; // > 0: the releasing handle has synchronised its own streams (IdleScope)
thread_local bool t_in_capture = false;
thread_local std::vector<std::pair<void*, size_t>> t_parked_dev, t_parked_pin;

bool disabled()
{
    static const bool off = std::getenv("T4A_NO_POOL") != nullptr;
    return off;
}

size_t class_of(size_t bytes)
{
    size_t c = 256;
    while (c < bytes) c <<= 1;
    return c;
}

int current_device()
{
    int dev = 0;
    T4A_HIP(hipGetDevice(&dev));
    return dev;
}

// device a block was allocated on (a process may drive several devices: the free list of the CURRENT device is the wrong one then)
int device_of(const void* p)
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) == hipSuccess) return attr.device;
    (void)hipGetLastError();
    return current_device();
}

// hipFree / hipHostFree synchronise the device: never while another thread of ours records a graph (g_capture_mu); a thread
// that is recording itself holds the mutex already and its callers park their blocks instead of coming here (ADVICE round 3:
// the trim loops and the over-limit frees did not take it)
template <class F> void free_outside_capture(F&& f)
{
    if (t_in_capture) {
        f();
        return;
    }
    std::lock_guard<std::mutex> lk(g_capture_mu);
    f();
}

// hands every cached device block of `dev` back to the driver (allocation failure: the cache itself may be what fills the HBM)
void trim_device_cache(int dev)
{
    std::vector<void*> blocks;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto& kv : g_dev[dev].dev_free) {
            for (void* b : kv.second) {
                blocks.push_back(b);
                g_cached_bytes -= kv.first;
            }
            kv.second.clear();
        }
    }
    free_outside_capture([&] {
        for (void* b : blocks) (void)hipFree(b);
    });
}

void trim_pin_cache()
{
    std::vector<void*> blocks;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto& kv : g_pin_free) {
            for (void* b : kv.second) blocks.push_back(b);
            kv.second.clear();
        }
        g_cached_pin_bytes = 0;
    }
    free_outside_capture([&] {
        for (void* b : blocks) (void)hipHostFree(b);
    });
}

// device-wide synchronisation outside any graph capture of ours; false when the device reported an (asynchronous) error —
// a block or stream that may still be in use must not be recycled then
bool quiesce_ok()
{
    std::lock_guard<std::mutex> lk(g_capture_mu);
    return hipDeviceSynchronize() == hipSuccess;
}

} // namespace

void quiesce() { (void)quiesce_ok(); }

void IdleScope::arm()
{
    if (!armed) {
        armed = true;
        ++t_idle_scope;
    }
}
IdleScope::~IdleScope()
{
    if (armed) --t_idle_scope;
}

void capture_begin()
{
    g_capture_mu.lock();
    t_in_capture = true;
}
void capture_end()
{
    t_in_capture = false;
    g_capture_mu.unlock();
    std::vector<std::pair<void*, size_t>> d, h;
    d.swap(t_parked_dev);
    h.swap(t_parked_pin);
    for (auto& b : d) dev_free(b.first, b.second);
    for (auto& b : h) pin_free(b.first, b.second);
}

void* dev_alloc(size_t bytes, size_t* got)
{
    if (bytes == 0) bytes = 1;
    if (disabled() || bytes >= kBypass) {
        void* p = nullptr;
        T4A_HIP(hipMalloc(&p, bytes));
        *got = bytes;
        return p;
    }
    const size_t c = class_of(bytes);
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto& fl = g_dev[dev].dev_free[c];
        if (!fl.empty()) {
            void* p = fl.back();
            fl.pop_back();
            g_cached_bytes -= c;
            *got = c;
            return p;
        }
    }
    void* p = nullptr;
    if (hipMalloc(&p, c) != hipSuccess) { // the cached blocks of other size classes may be what is in the way: release them, try once more
        (void)hipGetLastError();
        trim_device_cache(dev);
        T4A_HIP(hipMalloc(&p, c));
    }
    *got = c;
    return p;
}

void dev_free(void* p, size_t got)
{
    if (!p) return;
    if (t_in_capture) {
        t_parked_dev.emplace_back(p, got);
        return;
    }
    if (disabled() || got >= kBypass) {
        std::lock_guard<std::mutex> lk(g_capture_mu); // hipFree synchronises the device: never inside another thread's capture
        (void)hipFree(p);
        return;
    }
    // the block may still be in use by work in flight (hipFree would have waited for it) — unless its owner vouches for it
    const bool idle = t_idle_scope > 0 ? true : quiesce_ok();
    const int dev = device_of(p);
    if (idle) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_cached_bytes + got <= kCacheLimit) {
            g_dev[dev].dev_free[got].push_back(p);
            g_cached_bytes += got;
            return;
        }
    }
    free_outside_capture([&] { (void)hipFree(p); }); // over the limit, or the device is in an error state: not ours to hand out again
}

void* pin_alloc(size_t bytes, size_t* got)
{
    if (bytes == 0) bytes = 1;
    if (disabled() || bytes >= kBypass) {
        void* p = nullptr;
        T4A_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        *got = bytes;
        return p;
    }
    const size_t c = class_of(bytes);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto& fl = g_pin_free[c];
        if (!fl.empty()) {
            void* p = fl.back();
            fl.pop_back();
            g_cached_pin_bytes -= c;
            *got = c;
            return p;
        }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, c, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        trim_pin_cache();
        T4A_HIP(hipHostMalloc(&p, c, hipHostMallocDefault));
    }
    *got = c;
    return p;
}

void pin_free(void* p, size_t got)
{
    if (!p) return;
    if (t_in_capture) {
        t_parked_pin.emplace_back(p, got);
        return;
    }
    if (disabled() || got >= kBypass) {
        std::lock_guard<std::mutex> lk(g_capture_mu);
        (void)hipHostFree(p);
        return;
    }
    const bool idle = t_idle_scope > 0 ? true : quiesce_ok(); // a kernel may still be writing its result mirror / reading accumulators in place
    if (idle) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_cached_pin_bytes + got <= kPinCacheLimit) {
            g_pin_free[got].push_back(p);
            g_cached_pin_bytes += got;
            return;
        }
    }
    free_outside_capture([&] { (void)hipHostFree(p); });
}

hipStream_t stream_get(int kind)
{
    const int dev = current_device();
    if (!disabled()) {
        std::lock_guard<std::mutex> lk(g_mu);
        auto& v = g_dev[dev].streams[kind];
        if (!v.empty()) {
            hipStream_t s = v.back();
            v.pop_back();
            return s;
        }
    }
    hipStream_t s = nullptr;
    int least = 0, greatest = 0;
    T4A_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    static const bool flat = diag_env("T4A_FLAT_PRIORITY") != nullptr;
    if (kind == 0 || flat || least == greatest)
        T4A_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    else
        T4A_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, kind == 1 ? greatest : least));
    return s;
}

void stream_put(hipStream_t s, int kind)
{
    if (!s) return;
    const bool idle = hipStreamSynchronize(s) == hipSuccess;
    if (disabled() || !idle) { // (a stream whose work faulted is not handed to the next handle)
        (void)hipStreamDestroy(s);
        return;
    }
    const int dev = current_device();
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev[dev].streams[kind].push_back(s);
}

int compute_units()
{
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> lk(g_mu);
        const int c = g_dev[dev].cus;
        if (c > 0) return c;
    }
    hipDeviceProp_t prop;
    T4A_HIP(hipGetDeviceProperties(&prop, dev)); // (a millisecond: once per device, not once per handle)
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev[dev].cus = prop.multiProcessorCount;
    return prop.multiProcessorCount;
}

} // namespace pool
} // namespace t4a
