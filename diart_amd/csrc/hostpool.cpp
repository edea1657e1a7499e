#include "hostpool.h"

#include <stdlib.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr int MAX_WORKERS = 63;
// A worker polls this long for the next job before it sleeps: the two stages of one step arrive back to back, the
// next step ~1 ms later.  dz_host_pool_set_spin() overrides (0 = sleep at once): with 8 ranks on a node's 16 usable cores a
// rank has two cores for its launching thread and its pool, and a spinning worker takes one of them from the thread
// that feeds the GPU (StreamBatch sets it from the cores a rank really has).
std::atomic<int> g_spin_us{40};
int spin_us() { return g_spin_us.load(std::memory_order_relaxed); }

// One parallel-for.  Lives on the heap and is shared with the workers that picked it up, so that a worker which
// wakes up AFTER the job is complete (a sleeping thread can take a scheduler tick — 4 ms — to get a core on a busy
// node) finds every index handed out and walks away: the caller waits for the ITEMS, never for the workers.
// Until round 6 every spawned worker had to acknowledge every job; the launching thread then sat 4 - 8 ms in a
// 0.3 ms clustering / tail call about once per 200 steps while eight GPU steps drained (profiles/r06z_launch_stalls.json).
struct Job {
    const std::function<void(int, int)>* fn = nullptr;     // valid until done == n (the caller does not return before)
    int n = 0, helpers = 0;                                // helpers: workers 1..helpers may take indices
    std::atomic<int> next{0}, done{0};                     // next index to hand out; items completed
};

struct Pool {
    std::mutex callers;                       // one parallel-for at a time
    std::mutex mu;                            // guards cur / the sleeping workers
    std::condition_variable wake;
    std::atomic<unsigned long long> gen{0};   // bumped once per job
    std::shared_ptr<Job> cur;                 // the job in flight (null between jobs)
    int spawned = 0;
    pid_t owner = 0;
};

Pool* g_pool = nullptr;                       // leaked on purpose: workers outlive static destruction
std::mutex g_pool_mu;

void drain(Job* j, int worker) {
    for (;;) {
        const int i = j->next.fetch_add(1, std::memory_order_relaxed);
        if (i >= j->n) return;
        (*j->fn)(worker, i);
        j->done.fetch_add(1, std::memory_order_release);
    }
}

// `seen` starts at the generation current when the worker was spawned (read under `callers`, where gen cannot
// move): a worker never looks at a job published before it existed.
void worker_main(Pool* p, int id, unsigned long long seen) {
    for (;;) {
        // wait for a job newer than the last one this worker looked at: spin briefly, then sleep
        const auto t0 = std::chrono::steady_clock::now();
        while (p->gen.load(std::memory_order_acquire) == seen) {
            if (std::chrono::steady_clock::now() - t0 >= std::chrono::microseconds(spin_us())) {
                std::unique_lock<std::mutex> lk(p->mu);
                p->wake.wait(lk, [&] { return p->gen.load(std::memory_order_acquire) != seen; });
                break;
            }
            __builtin_ia32_pause();
        }
        std::shared_ptr<Job> j;
        {
            std::lock_guard<std::mutex> lk(p->mu);
            seen = p->gen.load(std::memory_order_acquire);
            j = p->cur;                        // null: that job is already complete
        }
        if (j && id <= j->helpers) drain(j.get(), id);
    }
}

Pool* pool() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    const pid_t me = getpid();
    if (!g_pool || g_pool->owner != me) {     // first use, or a forked child (threads do not survive fork)
        g_pool = new Pool();
        g_pool->owner = me;
    }
    return g_pool;
}

}  // namespace

void dz_host_parallel(int n, int threads, const std::function<void(int, int)>& fn) {
    if (n <= 0) return;
    if (threads > n) threads = n;
    if (threads > MAX_WORKERS + 1) threads = MAX_WORKERS + 1;
    if (threads <= 1) {
        for (int i = 0; i < n; ++i) fn(0, i);
        return;
    }
    Pool* p = pool();
    std::lock_guard<std::mutex> one(p->callers);
    while (p->spawned < threads - 1) {
        const int id = ++p->spawned;
        std::thread(worker_main, p, id, p->gen.load(std::memory_order_acquire)).detach();
    }
    auto job = std::make_shared<Job>();
    job->fn = &fn;
    job->n = n;
    job->helpers = threads - 1;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->cur = job;
        p->gen.fetch_add(1, std::memory_order_release);
    }
    p->wake.notify_all();
    drain(job.get(), 0);
    // wait for the items other threads are still inside: a short spin, then give the core away — on an
    // oversubscribed rank the worker that holds the last item may be waiting for exactly this core
    for (int spins = 0; job->done.load(std::memory_order_acquire) < n; ++spins) {
        if (spins < 2000)
            __builtin_ia32_pause();
        else
            std::this_thread::yield();
    }
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->cur.reset();
    }
}

extern "C" int dz_host_pool_set_spin(int microseconds) {
    g_spin_us.store(microseconds < 0 ? 0 : microseconds, std::memory_order_relaxed);
    return 0;
}
