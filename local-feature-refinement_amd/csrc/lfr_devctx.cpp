// Per-device streams + slab caches, pinned host arrays (see lfr_devctx.hpp).
#include "lfr_devctx.hpp"

#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <pthread.h>
#include <new>
#include <mutex>
#include <exception>
#include <thread>

#include "lfr_internal.hpp"

namespace lfr {

namespace {

constexpr size_t kPinThreshold = (size_t)1 << 20;      // smaller arrays: plain malloc (tests build thousands of toy graphs)

int hip_device_count() {                                // cached; 0 without a GPU (CPU-only test runs)
    static const int n = [] {
        if (const char *e = getenv("LFR_NO_PINNED")) if (e[0] == '1') return 0;
        int c = 0;
        if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); c = 0; }
        return c;
    }();
    return n;
}

size_t env_mb(const char *name, size_t dflt_bytes) {
    if (const char *e = getenv(name)) { const long long v = atoll(e); if (v >= 0) return (size_t)v << 20; }
    return dflt_bytes;
}

}  // namespace

hipError_t stream_wait(hipStream_t st) {
    static const bool spin = [] { const char *e = getenv("LFR_BLOCKING_SYNC"); return !(e && e[0] == '1'); }();
    if (spin) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0;; ++it) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) return q;
            if ((it & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
        }
    }
    return hipStreamSynchronize(st);
}

void *host_alloc(size_t bytes, bool *pinned, bool allow_pin) {
    *pinned = false;
    if (!allow_pin && bytes >= ((size_t)4 << 20)) {             // large and not to be pinned: transparent huge pages (THP runs in madvise mode here)
        void *p = nullptr;
        const size_t len = (bytes + (((size_t)2 << 20) - 1)) & ~(((size_t)2 << 20) - 1);
        if (posix_memalign(&p, (size_t)2 << 20, len) == 0 && p) { (void)madvise(p, len, MADV_HUGEPAGE); return p; }
    }
    if (allow_pin && bytes >= kPinThreshold && hip_device_count() > 0) {
        void *p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocPortable) == hipSuccess && p) { *pinned = true; return p; }
        (void)hipGetLastError();
    }
    void *p = malloc(bytes ? bytes : 1);
    if (!p) { fprintf(stderr, "lfr: out of host memory (%zu bytes)\n", bytes); abort(); }
    return p;
}

void host_free(void *p, bool pinned) {
    if (!p) return;
    if (pinned) (void)hipHostFree(p); else free(p);
}

// ------------------------------------------------------------------------------------------------
void *DevCtx::dev_acquire(size_t bytes, size_t *got) {
    bytes = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    {
        std::lock_guard<std::mutex> lk(mu);
        int best = -1;
        for (int i = 0; i < (int)free_dev.size(); ++i)
            if (free_dev[i].bytes >= bytes && free_dev[i].bytes <= 2 * bytes + ((size_t)64 << 20) &&
                (best < 0 || free_dev[i].bytes < free_dev[best].bytes)) best = i;
        if (best >= 0) {
            Slab s = free_dev[best];
            free_dev.erase(free_dev.begin() + best);
            cached_dev -= s.bytes;
            if (got) *got = s.bytes;
            return s.p;
        }
    }
    (void)hipSetDevice(device);
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        trim();                                         // cached slabs of the wrong size may be what is in the way
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc of %zu bytes failed on device %d", bytes, device); return nullptr; }
    }
    if (got) *got = bytes;
    return p;
}

void DevCtx::dev_release(void *p, size_t bytes) {
    if (!p) return;
    std::vector<Slab> drop;
    {
        std::lock_guard<std::mutex> lk(mu);
        free_dev.push_back(Slab{p, bytes});
        cached_dev += bytes;
        while (cached_dev > limit_dev && !free_dev.empty()) {       // oldest first
            drop.push_back(free_dev.front());
            cached_dev -= free_dev.front().bytes;
            free_dev.erase(free_dev.begin());
        }
    }
    if (!drop.empty()) { (void)hipSetDevice(device); for (auto &s : drop) (void)hipFree(s.p); }
}

void *DevCtx::pinned_acquire(size_t bytes, size_t *got) {
    bytes = (bytes + 4095) & ~(size_t)4095;
    {
        std::lock_guard<std::mutex> lk(mu);
        int best = -1;
        for (int i = 0; i < (int)free_pinned.size(); ++i)
            if (free_pinned[i].bytes >= bytes && free_pinned[i].bytes <= 2 * bytes + ((size_t)1 << 20) &&
                (best < 0 || free_pinned[i].bytes < free_pinned[best].bytes)) best = i;
        if (best >= 0) {
            Slab s = free_pinned[best];
            free_pinned.erase(free_pinned.begin() + best);
            cached_pinned -= s.bytes;
            if (got) *got = s.bytes;
            return s.p;
        }
    }
    (void)hipSetDevice(device);
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); set_error("hipHostMalloc of %zu bytes failed", bytes); return nullptr; }
    if (got) *got = bytes;
    return p;
}

void DevCtx::pinned_release(void *p, size_t bytes) {
    if (!p) return;
    std::vector<Slab> drop;
    {
        std::lock_guard<std::mutex> lk(mu);
        free_pinned.push_back(Slab{p, bytes});
        cached_pinned += bytes;
        while (cached_pinned > limit_pinned && !free_pinned.empty()) {
            drop.push_back(free_pinned.front());
            cached_pinned -= free_pinned.front().bytes;
            free_pinned.erase(free_pinned.begin());
        }
    }
    for (auto &s : drop) (void)hipHostFree(s.p);
}

hipStream_t DevCtx::side_stream(int i) {
    if (i < 0 || i >= kSideStreams) { set_error("side stream %d out of range", i); return nullptr; }
    std::lock_guard<std::mutex> lk(side_mu);
    if (!s_side[i]) {
        (void)hipSetDevice(device);
        if (hipStreamCreateWithFlags(&s_side[i], hipStreamNonBlocking) != hipSuccess) {
            set_error("hipStreamCreate failed on device %d: %s", device, hipGetErrorString(hipGetLastError()));
            s_side[i] = nullptr;
        }
    }
    return s_side[i];
}

hipEvent_t DevCtx::event_acquire(bool timing) {
    {
        std::lock_guard<std::mutex> lk(mu);
        std::vector<hipEvent_t> &pool = timing ? free_ev_timing : free_ev_plain;
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    }
    hipEvent_t e = nullptr;
    const hipError_t rc = timing ? hipEventCreate(&e) : hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) { set_error("hipEventCreate failed: %s", hipGetErrorString(rc)); return nullptr; }
    return e;
}
void DevCtx::event_release(hipEvent_t e, bool timing) {
    if (!e) return;
    std::lock_guard<std::mutex> lk(mu);
    std::vector<hipEvent_t> &pool = timing ? free_ev_timing : free_ev_plain;
    if (pool.size() < 4096) pool.push_back(e); else (void)hipEventDestroy(e);
}

void DevCtx::trim() {
    std::vector<Slab> d, h;
    {
        std::lock_guard<std::mutex> lk(mu);
        d.swap(free_dev); h.swap(free_pinned);
        cached_dev = cached_pinned = 0;
    }
    (void)hipSetDevice(device);
    for (auto &s : d) (void)hipFree(s.p);
    for (auto &s : h) (void)hipHostFree(s.p);
}

DevCtx *dev_ctx(int device) {
    static std::mutex mu;
    static std::vector<DevCtx *> ctxs;                   // never destroyed: the HIP runtime may be gone at exit
    std::lock_guard<std::mutex> lk(mu);
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess) { (void)hipGetLastError(); n_dev = 0; }
    if (device < 0 || device >= n_dev) { set_error("HIP device %d not available (%d devices)", device, n_dev); return nullptr; }
    if ((int)ctxs.size() <= device) ctxs.resize(device + 1, nullptr);
    if (ctxs[device]) return ctxs[device];
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed", device); return nullptr; }
    std::unique_ptr<DevCtx> c(new DevCtx());
    c->device = device;
    if (hipStreamCreateWithFlags(&c->s_main, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->s_copy, hipStreamNonBlocking) != hipSuccess) {
        set_error("hipStreamCreate failed on device %d: %s", device, hipGetErrorString(hipGetLastError()));
        return nullptr;
    }
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) c->n_cu = cu; else (void)hipGetLastError(); }
    c->limit_dev = env_mb("LFR_SLAB_CACHE_MB", c->limit_dev);
    c->limit_pinned = env_mb("LFR_PINNED_CACHE_MB", c->limit_pinned);
    ctxs[device] = c.release();
    return ctxs[device];
}

// ---------------------------------------------------------------------------- persistent host workers
namespace {
struct Pool {
    std::mutex mu, one;                         // `one`: one run_on_pool at a time
    std::condition_variable wake, done, task_done;
    const std::function<void()> *job = nullptr;
    int tickets = 0, running = 0, n_threads = 0;
    std::exception_ptr job_err;                 // first exception of a run_on_pool job on a worker (rethrown by run_on_pool)
    std::deque<std::shared_ptr<PoolTask>> tasks;
    void grow(int n) { while (n_threads < n) { std::thread(&Pool::worker, this).detach(); ++n_threads; } }     // (mu held)
    void run_task(std::unique_lock<std::mutex> &lk) {                                                             // (mu held, a task queued)
        std::shared_ptr<PoolTask> t = tasks.front();
        tasks.pop_front();
        lk.unlock();
        std::exception_ptr err;
        try { t->fn(); } catch (...) { err = std::current_exception(); }      // (never out of a detached worker; the waiter rethrows)
        lk.lock();
        t->err = err;
        t->done = true;
        task_done.notify_all();
    }
    void worker() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            wake.wait(lk, [&] { return tickets > 0 || !tasks.empty(); });
            if (!tasks.empty()) { run_task(lk); continue; }
            --tickets;
            const std::function<void()> *f = job;
            lk.unlock();
            std::exception_ptr err;
            try { (*f)(); } catch (...) { err = std::current_exception(); }
            lk.lock();
            if (err && !job_err) job_err = err;
            if (--running == 0) done.notify_all();
        }
    }
};
// never destroyed: its workers sleep until the process ends.  A forked child has none of the threads (and possibly a mutex that was
// held at the fork): it starts with a pool of its own (pthread_atfork; the parent's object is leaked in the child).
std::atomic<Pool *> g_pool{nullptr};
std::mutex g_pool_mu;
Pool *pool() {
    Pool *p = g_pool.load(std::memory_order_acquire);
    if (p) return p;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    p = g_pool.load(std::memory_order_relaxed);
    if (!p) {
        static bool registered = false;
        if (!registered) { registered = true; pthread_atfork(nullptr, nullptr, [] { g_pool.store(nullptr); new (&g_pool_mu) std::mutex(); }); }
        p = new Pool();
        g_pool.store(p, std::memory_order_release);
    }
    return p;
}
}  // namespace

void run_on_pool(int threads, const std::function<void()> &work) {
    if (threads <= 1) { work(); return; }
    Pool &P = *pool();
    std::unique_lock<std::mutex> only(P.one, std::try_to_lock);
    if (!only.owns_lock()) {                    // the pool is busy with another caller's work: threads of our own, as before
        std::vector<std::thread> th;
        for (int t = 1; t < threads; ++t) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        return;
    }
    {
        std::lock_guard<std::mutex> lk(P.mu);
        P.grow(threads - 1);
        P.job = &work; P.tickets = threads - 1; P.running = threads - 1;
    }
    P.wake.notify_all();
    std::exception_ptr err;
    try { work(); } catch (...) { err = std::current_exception(); }       // (the workers still hold `work`: wait for them before unwinding)
    std::unique_lock<std::mutex> lk(P.mu);
    P.done.wait(lk, [&] { return P.running == 0; });
    P.job = nullptr;
    if (!err) err = P.job_err;
    P.job_err = nullptr;
    lk.unlock();
    only.unlock();
    if (err) std::rethrow_exception(err);
}

std::shared_ptr<PoolTask> pool_async(std::function<void()> fn) {
    Pool &P = *pool();
    std::shared_ptr<PoolTask> t = std::make_shared<PoolTask>();
    t->fn = std::move(fn);
    {
        std::lock_guard<std::mutex> lk(P.mu);
        const unsigned hw = std::thread::hardware_concurrency();
        P.grow((int)std::min(64u, std::max(4u, hw / 2)));
        P.tasks.push_back(t);
    }
    P.wake.notify_one();
    return t;
}

void pool_wait(const std::shared_ptr<PoolTask> &t) {
    Pool &P = *pool();
    std::unique_lock<std::mutex> lk(P.mu);
    while (!t->done) {
        if (!P.tasks.empty()) P.run_task(lk);   // (a waiting thread works: tasks that spawn and wait for tasks cannot starve each other of workers)
        else P.task_done.wait(lk);
    }
    if (t->err) { const std::exception_ptr e = t->err; t->err = nullptr; lk.unlock(); std::rethrow_exception(e); }
}

}  // namespace lfr

extern "C" int64_t lfr_debug_pool_selftest(int threads, int64_t items, int reps) {
    int64_t done = 0;
    for (int r = 0; r < reps; ++r) {
        std::atomic<int64_t> next{0}, count{0};
        lfr::run_on_pool(threads, [&] {
            for (;;) {
                const int64_t i = next.fetch_add(1);
                if (i >= items) break;
                count.fetch_add(1);
            }
        });
        done += count.load();
    }
    return done;
}
