// common.hpp — shared host-side plumbing for the gfx950 backend (error channel, HIP checks, buffers).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/t4a_gpu.h"
#include "diag.hpp"

namespace t4a {

// Error carried up to the extern "C" boundary (mirrors CapiResult, tensor4all-capi/src/lib.rs:75).
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& msg);



inline void hip_check(hipError_t e, const char* what, const char* file, int line)
{
    if (e != hipSuccess) {
        char buf[512];
        std::snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
        int code = (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
                       ? T4A_GPU_NO_DEVICE
                       : T4A_GPU_INTERNAL_ERROR;
        throw Error(code, buf);
    }
}
#define T4A_HIP(expr) ::t4a::hip_check((expr), #expr, __FILE__, __LINE__)

// Fails loudly when there is no GPU: the product never falls back to a CPU path.
void require_device();

// process-wide caches of device blocks, pinned blocks and streams (pool.hip)
namespace pool {
void* dev_alloc(size_t bytes, size_t* got);
void dev_free(void* p, size_t got);
void* pin_alloc(size_t bytes, size_t* got);
void pin_free(void* p, size_t got);
hipStream_t stream_get(int kind); // 0 default priority, 1 highest, 2 lowest; non-blocking streams
void stream_put(hipStream_t s, int kind);
int compute_units();
// Declared as the FIRST member of a handle (destroyed last): once armed — by the handle's destructor, after it has synchronised
// every stream its buffers were ever used on — the buffers released on this thread until the scope ends go back to the cache
// without the device-wide synchronisation (which would wait for every other handle's work in flight).
struct IdleScope {
    bool armed = false;
    void arm();
    ~IdleScope();
};
void quiesce();       // device-wide synchronisation, never concurrent with a graph capture on one of our streams
void capture_begin(); // bracket hipStreamBeginCapture ... hipStreamEndCapture with these
void capture_end();
} // namespace pool

// Simple grow-only device buffer (blocks come from / return to the process-wide cache).
template <class T> struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    size_t bytes_ = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap), bytes_(o.bytes_)
    {
        o.p = nullptr;
        o.cap = 0;
        o.bytes_ = 0;
    }
    DevBuf& operator=(DevBuf&& o) noexcept
    {
        if (this != &o) {
            if (p) pool::dev_free(p, bytes_);
            p = o.p;
            cap = o.cap;
            bytes_ = o.bytes_;
            o.p = nullptr;
            o.cap = 0;
            o.bytes_ = 0;
        }
        return *this;
    }
    ~DevBuf()
    {
        if (p) pool::dev_free(p, bytes_);
    }
    void reserve(size_t n)
    {
        if (n <= cap) return;
        if (p) pool::dev_free(p, bytes_);
        p = nullptr;
        cap = 0;
        const size_t want = n + n / 4 + 64;
        p = static_cast<T*>(pool::dev_alloc(want * sizeof(T), &bytes_));
        cap = bytes_ / sizeof(T);
    }
    T* get() const { return p; }
};

// Pinned host staging buffer (grow-only).
template <class T> struct PinBuf {
    T* p = nullptr;
    size_t cap = 0;
    size_t bytes_ = 0;
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    ~PinBuf()
    {
        if (p) pool::pin_free(p, bytes_);
    }
    void reserve(size_t n)
    {
        if (n <= cap) return;
        if (p) pool::pin_free(p, bytes_);
        p = nullptr;
        cap = 0;
        const size_t want = n + n / 4 + 64;
        p = static_cast<T*>(pool::pin_alloc(want * sizeof(T), &bytes_));
        cap = bytes_ / sizeof(T);
    }
    T* get() const { return p; }
};

struct EventTimer {
    hipEvent_t a = nullptr, b = nullptr;
    void init()
    {
        if (!a) {
            T4A_HIP(hipEventCreate(&a));
            T4A_HIP(hipEventCreate(&b));
        }
    }
    ~EventTimer()
    {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

} // namespace t4a
