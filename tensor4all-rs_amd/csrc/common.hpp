// common.hpp — shared host-side plumbing for the gfx950 backend (error channel, HIP checks, buffers).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/t4a_gpu.h"

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

// Simple grow-only device buffer.
template <class T> struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap)
    {
        o.p = nullptr;
        o.cap = 0;
    }
    DevBuf& operator=(DevBuf&& o) noexcept
    {
        if (this != &o) {
            if (p) (void)hipFree(p);
            p = o.p;
            cap = o.cap;
            o.p = nullptr;
            o.cap = 0;
        }
        return *this;
    }
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    void reserve(size_t n)
    {
        if (n <= cap) return;
        if (p) T4A_HIP(hipFree(p));
        p = nullptr;
        size_t want = n + n / 4 + 64;
        T4A_HIP(hipMalloc(&p, want * sizeof(T)));
        cap = want;
    }
    T* get() const { return p; }
};

// Pinned host staging buffer (grow-only).
template <class T> struct PinBuf {
    T* p = nullptr;
    size_t cap = 0;
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    ~PinBuf()
    {
        if (p) (void)hipHostFree(p);
    }
    void reserve(size_t n)
    {
        if (n <= cap) return;
        if (p) T4A_HIP(hipHostFree(p));
        p = nullptr;
        size_t want = n + n / 4 + 64;
        T4A_HIP(hipHostMalloc(&p, want * sizeof(T), hipHostMallocDefault));
        cap = want;
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
