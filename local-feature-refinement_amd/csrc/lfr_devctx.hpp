// Per-device runtime state of liblfr_hip.so: streams, a cache of device slabs and pinned staging
// buffers, and the host-array type the match graph is parsed into.
//
// Why: the reference's "Total time" span (solve.cc:487-641) is ~0.5 ms of solve kernels plus whatever the
// runtime around them costs.  hipMalloc/hipFree (hipFree synchronises the device), pageable PCIe copies and
// stream synchronisations were 98 % of that span in round 1.  Everything a pipeline run needs is therefore
// carved out of a few cached slabs, the graph's big arrays live in pinned host memory so that the flows
// cross PCIe at link speed (asynchronously, beside the graph stage), and results come back through a
// pinned staging buffer.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

namespace lfr {

// ---- host arrays: pinned (hipHostMalloc, portable) when a HIP device exists and the array is large ----
void *host_alloc(size_t bytes, bool *pinned, bool allow_pin = true);
void host_free(void *p, bool pinned);

template <class T>
class HostBuf {                       // the subset of std::vector the ingest code uses; T trivially copyable
public:
    HostBuf() = default;
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    ~HostBuf() { if (p_) host_free(p_, pinned_); }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    bool pinned() const { return pinned_ && p_; }
    T *data() { return p_; }
    const T *data() const { return p_; }
    T &operator[](size_t i) { return p_[i]; }
    const T &operator[](size_t i) const { return p_[i]; }
    void reserve(size_t n) {
        if (n <= cap_) return;
        bool pin = false;
        T *q = (T *)host_alloc(n * sizeof(T), &pin, allow_pin_);
        if (n_) memcpy(q, p_, n_ * sizeof(T));
        if (p_) host_free(p_, pinned_);
        p_ = q; cap_ = n; pinned_ = pin;
    }
    void resize(size_t n) {           // new elements are zero (solve.cc:460-472: zero-initialised flow arrays)
        grow(n);
        if (n > n_) memset((void *)(p_ + n_), 0, (n - n_) * sizeof(T));
        n_ = n;
    }
    void push_back(const T &v) { grow(n_ + 1); p_[n_++] = v; }
    void clear() { n_ = 0; }
    // false: plain (huge-page) memory even for a large array - for data that is copied to HBM once, right away (ingest straight to a device)
    void allow_pinning(bool b) { allow_pin_ = b; }

private:
    void grow(size_t n) { if (n > cap_) reserve(n > 2 * cap_ ? (n < 1024 ? 1024 : n) : 2 * cap_); }
    T *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
    bool pinned_ = false, allow_pin_ = true;
};

// ---- per-device context ----
struct DevCtx {
    int device = -1;
    int n_cu = 256;                   // compute units of the device (sizes the persistent launches)
    hipStream_t s_main = nullptr;     // graph stage, assembly
    hipStream_t s_copy = nullptr;     // bulk H2D of the flows, beside s_main
    // streams the solve forks its concurrent kernel launches onto (hipStreamCreate costs milliseconds: a batch that created its
    // own five spent 18 ms of a 23 ms batch creation there); created on first use, shared by the batches of this device
    static constexpr int kSideStreams = 12;
    hipStream_t s_side[kSideStreams] = {nullptr};
    std::mutex side_mu;                // (not `mu`: a stream creation must not hold up slab acquisitions of a running pipeline)
    hipStream_t side_stream(int i);    // nullptr + set_error() on failure
    std::mutex mu;
    struct Slab { void *p; size_t bytes; };
    std::vector<Slab> free_dev, free_pinned;
    size_t cached_dev = 0, cached_pinned = 0;
    size_t limit_dev = (size_t)6 << 30, limit_pinned = (size_t)1 << 30;

    // device memory: a cached slab of at least `bytes` (hipMalloc on a miss); *got = its real size
    void *dev_acquire(size_t bytes, size_t *got);
    void dev_release(void *p, size_t bytes);
    void *pinned_acquire(size_t bytes, size_t *got);
    void pinned_release(void *p, size_t bytes);
    void trim();                       // hipFree / hipHostFree everything cached
    // events: a one-shot pipeline created ~25 of them per batch (a timing ring slot, fork/join) at 3-10 us apiece; they are pooled like
    // the slabs.  A released event must have completed (the batch synchronises its streams before it lets go).
    std::vector<hipEvent_t> free_ev_timing, free_ev_plain;
    hipEvent_t event_acquire(bool timing);          // nullptr + set_error() on failure
    void event_release(hipEvent_t e, bool timing);
};
// Wait for a stream the way a latency-bound pipeline wants it: poll hipStreamQuery for up to ~200 ms (the blocking
// hipStreamSynchronize was measured returning 20-30 ms after the GPU had finished when the process had just run many
// host threads - an interrupt-driven wake-up), then fall back to the blocking call.
hipError_t stream_wait(hipStream_t st);

// context of HIP device `device` (created on first use: hipSetDevice + two non-blocking streams);
// nullptr + set_error() when the device does not exist
DevCtx *dev_ctx(int device);

// bump allocator over one slab; the slab goes back to the context's cache when the arena dies
struct DevArena {
    DevCtx *ctx = nullptr;
    char *base = nullptr;
    size_t cap = 0, top = 0;
    DevArena() = default;
    DevArena(const DevArena &) = delete;
    DevArena &operator=(const DevArena &) = delete;
    ~DevArena() { reset(); }
    bool init(DevCtx *c, size_t bytes) {
        reset();
        ctx = c;
        base = (char *)c->dev_acquire(bytes, &cap);
        top = 0;
        return base != nullptr;
    }
    void reset() { if (base && ctx) ctx->dev_release(base, cap); base = nullptr; cap = top = 0; }
    void *take(size_t bytes) {
        const size_t b = (bytes + 255) & ~(size_t)255;
        if (!base || top + b > cap) return nullptr;
        void *p = base + top;
        top += b;
        return p;
    }
    template <class T> T *take_n(size_t n) { return (T *)take((n ? n : 1) * sizeof(T)); }
};

}  // namespace lfr
