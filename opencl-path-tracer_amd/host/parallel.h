// A few persistent worker threads for the host library's per-frame loops (Mesh::refit: smooth normals of ~40 k vertices per tick; the
// reference does this work single-threaded inside Assimp / refit_bvh.cpp, and at 4-6 ms per frame never needed more).  Threads are started at
// the first use and parked on a condition variable in between; parallelFor splits [0, count) into one contiguous range per thread, the
// caller takes the first one itself.  Ranges are fixed by (count, thread count) alone: results never depend on timing.
// A worker that has just finished a range polls for the next loop for ~100 us before it parks, and the caller polls as long for its workers before
// it blocks: a tree build is half a dozen short loops in a row (bins of the top levels, subtrees, smooth normals), and a futex sleep + wake per loop
// and per thread cost more than the loops' own work.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <vector>

namespace raytracer {

class WorkerPool {
public:
    static WorkerPool& get()
    {
        // on the heap, never destroyed by a static destructor: a forked child inherits the object but not the workers, and glibc's
        // pthread_cond_destroy waits for the waiters the copied condition variable still counts -- forever.  The process that made the pool
        // stops its workers at exit (Stopper); a child leaves the copy alone.
        static WorkerPool* pool = new WorkerPool;
        static Stopper stopper { pool };
        (void)stopper;
        return *pool; // (never deleted: a static destructor that runs later may still come here -- see Stopper)
    }
    // fn(begin, end) on disjoint ranges covering [0, count); below `minPerThread` items per thread the caller does it alone
    void parallelFor(size_t count, size_t minPerThread, const std::function<void(size_t, size_t)>& fn)
    {
        const size_t parts = std::max<size_t>(1, std::min<size_t>(m_threads.size() + 1, count / std::max<size_t>(minPerThread, 1)));
        // (a loop inside a task of the pool runs where it is: the pool serves one parallelFor at a time; so does everything in a process that was
        // forked after the pool had started -- fork copies the calling thread only, the workers are not there)
        if (parts <= 1 || insideTask() || getpid() != m_pid) {
            if (count)
                fn(0, count);
            return;
        }
        std::unique_lock<std::mutex> callers(m_callers); // one parallelFor at a time
        const size_t chunk = (count + parts - 1) / parts;
        {
            std::lock_guard<std::mutex> lock(m_mutex);
            m_fn = &fn, m_count = count, m_chunk = chunk, m_parts = parts, m_pending.store(parts - 1, std::memory_order_relaxed);
            m_generation.fetch_add(1, std::memory_order_release);
        }
        m_wake.notify_all();
        // An exception out of fn -- on this thread or on a worker (std::bad_alloc from a builder's vectors) -- is kept, every range is still waited
        // for (the workers hold a pointer to the caller's `fn`, which captures the caller's stack), and the first one is rethrown HERE, on the
        // calling thread, where the C ABI's guarded() turns it into an error code.
        std::exception_ptr mine;
        insideTask() = true;
        try {
            fn(0, std::min(chunk, count));
        } catch (...) {
            mine = std::current_exception();
        }
        insideTask() = false;
        spinUntil([&] { return m_pending.load(std::memory_order_acquire) == 0; });
        std::unique_lock<std::mutex> lock(m_mutex);
        m_done.wait(lock, [&] { return m_pending.load(std::memory_order_relaxed) == 0; });
        m_fn = nullptr;
        std::exception_ptr first = mine ? mine : m_error;
        m_error = nullptr;
        lock.unlock();
        if (first)
            std::rethrow_exception(first);
    }
    size_t threads() const { return getpid() == m_pid ? m_threads.size() + 1 : 1; }
    static bool& insideTask()
    {
        static thread_local bool inside = false;
        return inside;
    }

private:
    // PTAMD_HOST_SPIN_US: how long a worker polls for the next loop before it parks (default 100; 0: park at once).  A frame loop whose ticks are a few
    // milliseconds apart finds its workers asleep at every tick: measured in EXPERIMENTS.md (round 6), not made the default -- seven cores polling for a whole
    // tick is the application's call, not a library's.
    static long spinMicroseconds()
    {
        static const long us = [] {
            const char* env = std::getenv("PTAMD_HOST_SPIN_US");
            return env ? std::clamp(std::atol(env), 0L, 1000000L) : 100L;
        }();
        return us;
    }
    // polls `ready` for about that long (the pause instruction between polls; the clock read every 64th)
    template <typename F>
    static bool spinUntil(F ready)
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned i = 1;; i++) {
            if (ready())
                return true;
            __builtin_ia32_pause();
            if ((i & 63u) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spinMicroseconds()))
                return false;
        }
    }
    WorkerPool()
    {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        unsigned want = 8u; // the caller is the first worker; a rebuilt tree per frame (16 subtrees of a 20 k-triangle mesh) is the loop that uses them all
        if (const char* env = std::getenv("PTAMD_HOST_THREADS"))
            want = (unsigned)std::clamp(std::atoi(env), 1, 64);
        const unsigned n = std::min(want, hw) - 1u;
        for (unsigned i = 0; i < n; i++)
            m_threads.emplace_back([this, i] { run(i + 1); });
    }
    // at exit the process that made the pool stops and joins its workers; the object itself stays (a static destructor that runs after this one may
    // still call get(): it finds a pool without workers, whose loops run on the caller)
    struct Stopper {
        WorkerPool* pool;
        ~Stopper()
        {
            if (getpid() == pool->m_pid)
                pool->stop();
        }
    };
    void stop()
    {
        {
            std::lock_guard<std::mutex> lock(m_mutex);
            m_quit = true;
        }
        m_wake.notify_all();
        for (std::thread& t : m_threads)
            t.join();
        m_threads.clear();
    }
    ~WorkerPool() = delete; // (see get())
    void run(size_t part)
    {
        size_t seen = 0;
        for (;;) {
            const std::function<void(size_t, size_t)>* fn;
            size_t begin, end;
            if (seen != 0)
                spinUntil([&] { return m_generation.load(std::memory_order_acquire) != seen; }); // (m_quit is only ever set while the workers are parked or will park)
            {
                std::unique_lock<std::mutex> lock(m_mutex);
                m_wake.wait(lock, [&] { return m_quit || m_generation.load(std::memory_order_relaxed) != seen; });
                if (m_quit)
                    return;
                seen = m_generation.load(std::memory_order_relaxed);
                if (part >= m_parts)
                    continue; // fewer parts than threads this time
                fn = m_fn, begin = std::min(part * m_chunk, m_count), end = std::min(begin + m_chunk, m_count);
            }
            std::exception_ptr err;
            if (begin < end) {
                insideTask() = true;
                try {
                    (*fn)(begin, end);
                } catch (...) {
                    err = std::current_exception();
                }
                insideTask() = false;
            }
            {
                std::lock_guard<std::mutex> lock(m_mutex);
                if (err && !m_error)
                    m_error = err;
                if (m_pending.fetch_sub(1, std::memory_order_acq_rel) == 1)
                    m_done.notify_one();
            }
        }
    }
    const pid_t m_pid = getpid();
    std::vector<std::thread> m_threads;
    std::mutex m_mutex, m_callers;
    std::condition_variable m_wake, m_done;
    const std::function<void(size_t, size_t)>* m_fn = nullptr;
    size_t m_count = 0, m_chunk = 0, m_parts = 0;
    std::atomic<size_t> m_pending { 0 }, m_generation { 0 };
    bool m_quit = false;
    std::exception_ptr m_error; // the first exception a worker's range threw in the current loop
};

} // namespace raytracer
