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
size_t g_cached_bytes = 0;
constexpr size_t kBypass = (size_t)256 << 20;  // blocks of this size and more go straight to the driver
constexpr size_t kCacheLimit = (size_t)16 << 30;

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

} // namespace

void quiesce()
{
    std::lock_guard<std::mutex> lk(g_capture_mu);
    (void)hipDeviceSynchronize();
}

void capture_begin() { g_capture_mu.lock(); }
void capture_end() { g_capture_mu.unlock(); }

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
    T4A_HIP(hipMalloc(&p, c));
    *got = c;
    return p;
}

void dev_free(void* p, size_t got)
{
    if (!p) return;
    if (disabled() || got >= kBypass) {
        (void)hipFree(p); // (synchronises the device)
        return;
    }
    // the block may still be in use by work in flight (hipFree would have waited for it)
    quiesce();
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_cached_bytes + got <= kCacheLimit) {
            g_dev[dev].dev_free[got].push_back(p);
            g_cached_bytes += got;
            return;
        }
    }
    (void)hipFree(p);
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
            *got = c;
            return p;
        }
    }
    void* p = nullptr;
    T4A_HIP(hipHostMalloc(&p, c, hipHostMallocDefault));
    *got = c;
    return p;
}

void pin_free(void* p, size_t got)
{
    if (!p) return;
    if (disabled() || got >= kBypass) {
        (void)hipHostFree(p);
        return;
    }
    quiesce(); // a kernel may still be writing its result mirror / reading accumulators in place
    std::lock_guard<std::mutex> lk(g_mu);
    g_pin_free[got].push_back(p);
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
    static const bool flat = std::getenv("T4A_FLAT_PRIORITY") != nullptr;
    if (kind == 0 || flat || least == greatest)
        T4A_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    else
        T4A_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, kind == 1 ? greatest : least));
    return s;
}

void stream_put(hipStream_t s, int kind)
{
    if (!s) return;
    (void)hipStreamSynchronize(s);
    if (disabled()) {
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
