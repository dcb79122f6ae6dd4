// A few persistent worker threads for the host library's per-frame loops (Mesh::refit: smooth normals of ~40 k vertices per tick; the
// reference does this work single-threaded inside Assimp / refit_bvh.cpp, and at 4-6 ms per frame never needed more).  Threads are started at
// the first use and parked on a condition variable in between; parallelFor splits [0, count) into contiguous ranges (up to four per thread) that
// the caller and the workers take one after the other from a shared counter.  The ranges are fixed by (count, minPerThread, thread count) alone:
// results never depend on timing; who runs which range does, and a loop never waits for a worker that has not turned up.
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
#include <pthread.h>
#include <sched.h>
#include <string>
#include <cstdio>
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
    // fn(begin, end) on disjoint ranges covering [0, count); below `minPerThread` items per thread the caller does it alone.  The ranges -- up to four per
    // thread -- are fixed by (count, minPerThread, thread count); WHO runs a range is not: every participant takes the next one from a shared counter, the
    // caller included, and the loop is over when every range has been run, not when every worker has reported.  A worker that wakes late (a core in a deep
    // sleep state, a neighbour's process on its core) finds nothing left and costs nothing; the caller alone would finish the loop.
    void parallelFor(size_t count, size_t minPerThread, const std::function<void(size_t, size_t)>& fn)
    {
        const size_t width = std::max<size_t>(1, std::min<size_t>(m_threads.size() + 1, count / std::max<size_t>(minPerThread, 1)));
        // (a loop inside a task of the pool runs where it is: the pool serves one parallelFor at a time; so does everything in a process that was
        // forked after the pool had started -- fork copies the calling thread only, the workers are not there)
        if (width <= 1 || insideTask() || getpid() != m_pid) {
            if (count)
                fn(0, count);
            return;
        }
        const size_t chunks = std::min<size_t>(count, std::min<size_t>(width * 4, std::max<size_t>(width, count / std::max<size_t>(minPerThread, 1))));
        std::unique_lock<std::mutex> callers(m_callers); // one parallelFor at a time
        uint64_t ticket;
        {
            std::lock_guard<std::mutex> lock(m_mutex);
            m_fn = &fn, m_count = count, m_chunks = chunks, m_chunk = (count + chunks - 1) / chunks;
            m_done.store(0, std::memory_order_relaxed);
            ticket = (uint64_t)(m_generation.load(std::memory_order_relaxed) + 1) << 32;
            m_next.store(ticket, std::memory_order_relaxed); // (generation, next range): a worker still holding an earlier loop's ticket can never take a range of this one
            m_generation.fetch_add(1, std::memory_order_release);
        }
        m_wake.notify_all();
        // An exception out of fn -- on this thread or on a worker (std::bad_alloc from a builder's vectors) -- is kept, every range is still run to its end
        // (the workers hold a pointer to the caller's `fn`, which captures the caller's stack), and the first one is rethrown HERE, on the calling
        // thread, where the C ABI's guarded() turns it into an error code.
        insideTask() = true;
        runRanges(ticket, &fn, count, chunks, m_chunk);
        insideTask() = false;
        spinUntil([&] { return m_done.load(std::memory_order_acquire) == chunks; });
        std::unique_lock<std::mutex> lock(m_mutex);
        m_finished.wait(lock, [&] { return m_done.load(std::memory_order_acquire) == chunks; });
        m_fn = nullptr;
        std::exception_ptr first = m_error;
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
            m_threads.emplace_back([this] { run(); });
        pinWorkers();
    }
    // PTAMD_HOST_PIN=l3 | numa | l3all: the workers (l3all: and the thread that makes the pool) are restricted to the CPUs that share a last-level cache
    // (numa: a memory node) with the CPU the pool is made on.  On a two-socket host with sixteen L3 domains the scheduler otherwise spreads eight threads
    // of one loop over domains and sockets as it sees fit (measured: EXPERIMENTS.md, round 6).  Off by default: where threads run is the application's call.
    static std::vector<int> cpuList(const std::string& path)
    {
        std::vector<int> cpus;
        FILE* f = std::fopen(path.c_str(), "r");
        if (!f)
            return cpus;
        char buf[4096];
        if (std::fgets(buf, sizeof buf, f))
            for (char* p = buf; *p;) {
                char* e;
                const long a = std::strtol(p, &e, 10);
                if (e == p)
                    break;
                long b = a;
                if (*e == '-')
                    b = std::strtol(e + 1, &e, 10);
                for (long c = a; c <= b && c < CPU_SETSIZE; c++)
                    cpus.push_back((int)c);
                p = *e == ',' ? e + 1 : e;
                if (*e != ',')
                    break;
            }
        std::fclose(f);
        return cpus;
    }
    void pinWorkers()
    {
        const char* mode = std::getenv("PTAMD_HOST_PIN");
        if (!mode || !*mode || std::string(mode) == "off")
            return;
        const int cpu = sched_getcpu();
        if (cpu < 0)
            return;
        std::vector<int> cpus;
        const std::string base = "/sys/devices/system/cpu/cpu" + std::to_string(cpu);
        if (std::string(mode) == "numa") {
            for (int node = 0; node < 64 && cpus.empty(); node++) {
                const std::vector<int> of = cpuList("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
                if (std::find(of.begin(), of.end(), cpu) != of.end())
                    cpus = of;
            }
        } else {
            cpus = cpuList(base + "/cache/index3/shared_cpu_list");
        }
        if (cpus.empty())
            return;
        cpu_set_t set;
        CPU_ZERO(&set);
        for (int c : cpus)
            CPU_SET(c, &set);
        for (std::thread& t : m_threads)
            (void)pthread_setaffinity_np(t.native_handle(), sizeof set, &set);
        if (std::string(mode) == "l3all")
            (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
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
    // takes ranges of the loop `ticket` belongs to until none is left; every range that was taken is counted when it is over, whatever it threw
    void runRanges(uint64_t ticket, const std::function<void(size_t, size_t)>* fn, size_t count, size_t chunks, size_t chunk)
    {
        for (;;) {
            uint64_t v = m_next.load(std::memory_order_acquire);
            if ((v >> 32) != (ticket >> 32) || (v & 0xFFFFFFFFu) >= chunks)
                return; // another loop's counter by now, or nothing left
            if (!m_next.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel))
                continue;
            const size_t i = (size_t)(v & 0xFFFFFFFFu), begin = std::min(i * chunk, count), end = std::min(begin + chunk, count);
            std::exception_ptr err;
            if (begin < end) {
                try {
                    (*fn)(begin, end);
                } catch (...) {
                    err = std::current_exception();
                }
            }
            if (err) {
                std::lock_guard<std::mutex> lock(m_mutex);
                if (!m_error)
                    m_error = err;
            }
            if (m_done.fetch_add(1, std::memory_order_acq_rel) + 1 == chunks) {
                std::lock_guard<std::mutex> lock(m_mutex); // (the caller may be between its check and its wait)
                m_finished.notify_one();
            }
        }
    }
    void run()
    {
        size_t seen = 0;
        for (;;) {
            const std::function<void(size_t, size_t)>* fn;
            size_t count, chunks, chunk;
            uint64_t ticket;
            if (seen != 0)
                spinUntil([&] { return m_generation.load(std::memory_order_acquire) != seen; }); // (m_quit is only ever set while the workers are parked or will park)
            {
                std::unique_lock<std::mutex> lock(m_mutex);
                m_wake.wait(lock, [&] { return m_quit || m_generation.load(std::memory_order_relaxed) != seen; });
                if (m_quit)
                    return;
                seen = m_generation.load(std::memory_order_relaxed);
                fn = m_fn, count = m_count, chunks = m_chunks, chunk = m_chunk, ticket = (uint64_t)seen << 32;
            }
            if (!fn)
                continue; // the loop is over already
            insideTask() = true;
            runRanges(ticket, fn, count, chunks, chunk);
            insideTask() = false;
        }
    }
    const pid_t m_pid = getpid();
    std::vector<std::thread> m_threads;
    std::mutex m_mutex, m_callers;
    std::condition_variable m_wake, m_finished;
    const std::function<void(size_t, size_t)>* m_fn = nullptr;
    size_t m_count = 0, m_chunk = 0, m_chunks = 0;
    std::atomic<size_t> m_done { 0 }, m_generation { 0 };
    std::atomic<uint64_t> m_next { 0 };
    bool m_quit = false;
    std::exception_ptr m_error; // the first exception a worker's range threw in the current loop
};

} // namespace raytracer
