#include "hostpool.h"

#include <stdlib.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
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

struct Pool {
    std::mutex callers;                       // one parallel-for at a time
    std::mutex mu;                            // guards gen / the sleeping workers
    std::condition_variable wake;
    std::atomic<unsigned long long> gen{0};   // bumped once per job
    // the job
    const std::function<void(int, int)>* fn = nullptr;
    int n = 0, helpers = 0;                   // helpers: workers 1..helpers may join this job
    std::atomic<int> next{0}, left{0};        // next index to hand out; workers that have not acknowledged
    int spawned = 0;
    pid_t owner = 0;
};

Pool* g_pool = nullptr;                       // leaked on purpose: workers outlive static destruction
std::mutex g_pool_mu;

void drain(Pool* p, int worker) {
    for (;;) {
        const int i = p->next.fetch_add(1, std::memory_order_relaxed);
        if (i >= p->n) return;
        (*p->fn)(worker, i);
    }
}

// `seen` starts at the generation current when the worker was spawned (read under `callers`, where
// gen cannot move): a worker never looks at — or acknowledges — a job published before it existed.
// Starting every worker at 0 let one spawned into a pool with gen > 0 acknowledge the job that was
// about to be published, and then that job again (`left` ended one short, the caller
// returned while a worker was still inside fn()).
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
        seen = p->gen.load(std::memory_order_acquire);
        if (id <= p->helpers) drain(p, id);
        p->left.fetch_sub(1, std::memory_order_acq_rel);   // every worker acknowledges every job
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
    // every spawned worker looks at and acknowledges every job (so the job fields are never rewritten
    // while a worker may still read them); only the first `helpers` of them take indices
    p->fn = &fn;
    p->n = n;
    p->helpers = threads - 1;
    p->next.store(0, std::memory_order_relaxed);
    p->left.store(p->spawned, std::memory_order_relaxed);
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->gen.fetch_add(1, std::memory_order_release);
    }
    p->wake.notify_all();
    drain(p, 0);
    // wait for the acknowledgements: a short spin, then give the core away — on an oversubscribed rank the worker
    // that has not acknowledged yet may be waiting for exactly this core
    for (int spins = 0; p->left.load(std::memory_order_acquire) > 0; ++spins) {
        if (spins < 2000)
            __builtin_ia32_pause();
        else
            std::this_thread::yield();
    }
    p->fn = nullptr;
}

extern "C" int dz_host_pool_set_spin(int microseconds) {
    g_spin_us.store(microseconds < 0 ? 0 : microseconds, std::memory_order_relaxed);
    return 0;
}
