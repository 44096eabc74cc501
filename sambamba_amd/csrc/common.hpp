// common.hpp -- shared host-side helpers for libsbx_depth (HIP error handling, device buffers).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <ctime>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sbx_depth.h"

namespace sbx {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define SBX_HIP(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            throw ::sbx::Error(SBX_ENODEVICE, std::string("HIP error: ") + hipGetErrorString(_e) +  \
                                                  " at " + __FILE__ + ":" + std::to_string(__LINE__)); \
    } while (0)

// RAII device allocation
// wall clock spent in hipMalloc / hipFree (SBX_TIMING reports it: device memory that another process has just freed is
// scrubbed by the driver before it is handed out again, which can dominate a one-shot run)
inline double& alloc_seconds() { static double s = 0; return s; }
inline double wall_now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

// growth slack of device buffers in percent (DevBuf::ensure); read and raised from several threads (sbx_prefetch_interval runs next to
// the thread that computes): an atomic
inline std::atomic<int>& devbuf_slack_pct() { static std::atomic<int> p{0}; return p; }

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() {}
    explicit DevBuf(size_t count) { alloc(count); }
    ~DevBuf() { release(); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    void alloc(size_t count) {
        release();
        n = count;
        if (count) {
            const double t0 = wall_now();
            hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
            alloc_seconds() += wall_now() - t0;
            if (e != hipSuccess) {
                p = nullptr; n = 0;
                throw Error(e == hipErrorOutOfMemory ? SBX_ENOMEM : SBX_ENODEVICE,
                            std::string("hipMalloc of ") + std::to_string(count * sizeof(T)) + " bytes failed: " + hipGetErrorString(e));
            }
        }
    }
    // grow-only (keeps the allocation when it is already large enough).  A buffer that has to grow takes `slack_pct` percent
    // more than asked for: consecutive runs of similar size (the slices of a pipelined job, the batches of a genome) then
    // keep their allocations -- hipFree synchronises the whole device, which stalls every other stream.
    void ensure(size_t count) { if (count > n) alloc(count + (size_t)((double)count * devbuf_slack_pct().load(std::memory_order_relaxed) / 100.0)); }
    void release() {
        if (p) { const double t0 = wall_now(); (void)hipFree(p); alloc_seconds() += wall_now() - t0; }
        p = nullptr; n = 0;
    }
    size_t bytes() const { return n * sizeof(T); }
};

// Device/stream bring-up; throws SBX_ENODEVICE when there is no usable GPU.
void require_device(int device);

struct EventTimer {
    hipEvent_t a = nullptr, b = nullptr;
    EventTimer() { SBX_HIP(hipEventCreate(&a)); SBX_HIP(hipEventCreate(&b)); }
    ~EventTimer() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    void start(hipStream_t s) { SBX_HIP(hipEventRecord(a, s)); }
    void stop(hipStream_t s) { SBX_HIP(hipEventRecord(b, s)); }
    double ms() { float f = 0; SBX_HIP(hipEventSynchronize(b)); SBX_HIP(hipEventElapsedTime(&f, a, b)); return (double)f; }
};

}  // namespace sbx
