// Sanitizer harness for the host half of the path (csrc/cluster.cpp, tail.cpp, hostpool.cpp, filebatch.cpp): the fp64
// OnlineSpeakerClustering, DelayedAggregation + Binarize and the worker pool that runs them for N streams.  Built by
// tests/test_host_sanitizers.py with g++ -fsanitize=address,undefined and with -fsanitize=thread (the GPU build cannot
// carry sanitizers on this pool) and run on random inputs — NaN embeddings, silent chunks, more local speakers than free
// centroids, varying thread counts, two callers at once.  It checks only what must hold whatever the data (finite
// scores, assignments in range, batch == one-by-one); the reference's results are pinned by tests/test_clustering.py and
// tests/test_tail.py.  Exit code 0 = nothing reported.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../../include/diart_amd.h"

// the two symbols the HIP translation units provide in the product library
static thread_local char g_err[512];
void dz_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* dz_last_error(void) { return g_err; }

#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) {                                                                \
            fprintf(stderr, "host_sanitize: %s failed at line %d (%s)\n", #cond, __LINE__, g_err); \
            exit(3);                                                                  \
        }                                                                             \
    } while (0)

namespace {
constexpr int F = 293, K = 3, D = 64, G = 20;

struct Streams {
    int n;
    std::vector<dz_clu*> clu;
    std::vector<dz_tail*> tail;
    explicit Streams(int n_) : n(n_), clu(n_), tail(n_) {
        std::vector<double> ham(F);
        for (int i = 0; i < F; ++i) ham[i] = 0.54 - 0.46 * std::cos(2.0 * M_PI * i / (F - 1));
        for (int i = 0; i < n; ++i) {
            CHECK(dz_clu_create(0.5, 0.3, 1.0, G, &clu[i]) == 0);
            CHECK(dz_tail_create(F, G, 0.5, 2.5, 0.5, DZ_AGG_HAMMING, DZ_CROP_LOOSE, ham.data(), &tail[i]) == 0);
        }
    }
    ~Streams() {
        for (int i = 0; i < n; ++i) {
            dz_clu_destroy(clu[i]);
            dz_tail_destroy(tail[i]);
        }
    }
};

void fill(std::mt19937& rng, int n, int step, std::vector<float>& seg, std::vector<float>& emb) {
    std::uniform_real_distribution<float> u(0.f, 1.f);
    std::normal_distribution<float> g(0.f, 1.f);
    seg.resize((size_t)n * F * K);
    emb.resize((size_t)n * K * D);
    for (int i = 0; i < n; ++i) {
        const int kind = (int)(rng() % 8);               // 0: silent chunk, 1: a NaN embedding, else speech
        for (int f = 0; f < F; ++f)
            for (int k = 0; k < K; ++k) {
                float v = u(rng);
                if (kind == 0) v *= 0.05f;
                else if ((f / 40 + k + step) % 3 == 0) v = 0.6f + 0.4f * v;      // active runs
                seg[((size_t)i * F + f) * K + k] = v;
            }
        for (int k = 0; k < K; ++k) {
            const int spk = (int)(rng() % 30);           // up to 30 identities for 20 centroids
            for (int d = 0; d < D; ++d)
                emb[((size_t)i * K + k) * D + d] = std::sin(0.37f * (float)(spk + 1) * (float)(d + 1)) + 0.05f * g(rng);
        }
        if (kind == 1) emb[((size_t)i * K + 1) * D + 7] = NAN;
    }
}

void run_streams(unsigned seed, int n, int steps, int threads) {
    std::mt19937 rng(seed);
    Streams s(n), one(n);
    std::vector<float> seg, emb;
    std::vector<double> scores((size_t)n * F * G), scores1((size_t)F * G), agg((size_t)n * (F + 2) * G), turns((size_t)n * 256 * 3);
    std::vector<double> start(n), res(n, 5.0 / F), t0(n), rout(n);
    std::vector<int> assign((size_t)n * K), assign1(K), rows(n), nturns(n);
    for (int t = 0; t < steps; ++t) {
        fill(rng, n, t, seg, emb);
        CHECK(dz_clu_step_batch(s.clu.data(), n, seg.data(), F, K, emb.data(), D, scores.data(), assign.data(), threads) == 0);
        for (int i = 0; i < n; ++i) {                    // the batch on the pool == one stream at a time on this thread
            CHECK(dz_clu_step(one.clu[i], &seg[(size_t)i * F * K], F, K, &emb[(size_t)i * K * D], D, scores1.data(), assign1.data()) == 0);
            CHECK(memcmp(scores1.data(), &scores[(size_t)i * F * G], sizeof(double) * F * G) == 0);
            CHECK(memcmp(assign1.data(), &assign[(size_t)i * K], sizeof(int) * K) == 0);
            for (int k = 0; k < K; ++k) CHECK(assign[(size_t)i * K + k] >= -1 && assign[(size_t)i * K + k] < G);
            start[i] = 0.5 * t;
        }
        for (double v : scores) CHECK(std::isfinite(v));
        CHECK(dz_tail_step_batch(s.tail.data(), n, scores.data(), start.data(), res.data(), agg.data(), rows.data(), t0.data(),
                                 rout.data(), turns.data(), 256, nturns.data(), threads) == 0);
        for (int i = 0; i < n; ++i) CHECK(rows[i] >= 0 && rows[i] <= F + 2 && nturns[i] >= 0 && nturns[i] <= 256);
        if (t % 17 == 16) {                              // a stream that ends and starts again
            CHECK(dz_clu_reset(s.clu[t % n]) == 0 && dz_clu_reset(one.clu[t % n]) == 0);
            CHECK(dz_tail_reset(s.tail[t % n]) == 0);
        }
    }
    std::vector<double> centers((size_t)G * D);
    std::vector<int> mask(G);
    for (int i = 0; i < n; ++i) {
        const int rc = dz_clu_get_centers(s.clu[i], centers.data(), D);
        CHECK(rc == 0 || rc == 1);
        CHECK(dz_clu_get_active(s.clu[i], mask.data()) == 0);
    }
}

void run_files(unsigned seed, int files, int steps, int threads) {
    std::mt19937 rng(seed);
    Streams s(files);
    const int per = 4, rows = files * per;
    std::vector<float> seg, emb;
    std::vector<int> row0(files), count(files), nturns(rows), assign((size_t)rows * K);
    std::vector<double> start(rows), turns((size_t)rows * 256 * 3);
    for (int t = 0; t < steps; ++t) {
        fill(rng, rows, t, seg, emb);
        for (int i = 0; i < files; ++i) {
            row0[i] = i * per;
            count[i] = 1 + (int)(rng() % per);
            for (int j = 0; j < per; ++j) start[(size_t)i * per + j] = 0.5 * (t * per + j);
        }
        CHECK(dz_file_step_batch(s.clu.data(), s.tail.data(), files, row0.data(), count.data(), seg.data(), F, K, emb.data(), D, G,
                                 start.data(), 5.0 / F, turns.data(), 256, nturns.data(), assign.data(), threads) == 0);
    }
}

void run_lsap(unsigned seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> u(0.0, 2.0);
    for (int it = 0; it < 300; ++it) {
        const int nr = 1 + (int)(rng() % 6), nc = 1 + (int)(rng() % 24);
        std::vector<double> cost((size_t)nr * nc);
        for (double& c : cost) c = rng() % 5 == 0 ? 1e10 : u(rng);
        std::vector<int> col(nr, -2);
        CHECK(dz_lsap(cost.data(), nr, nc, col.data()) == 0);
        for (int r = 0; r < nr; ++r) CHECK(col[r] >= -1 && col[r] < nc);
    }
}
}  // namespace

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 60;
    run_lsap(1);
    for (int threads : {1, 3, 8}) {
        run_streams(10u + threads, 16, steps, threads);
        run_files(20u + threads, 5, steps / 2, threads);
    }
    // two engines' host halves at once (two StreamBatch objects on two Python threads share the process-wide pool)
    dz_host_pool_set_spin(40);
    std::thread a([&] { run_streams(31, 12, steps, 4); }), b([&] { run_streams(32, 9, steps, 4); });
    std::thread c([&] { run_files(33, 4, steps / 2, 3); });
    a.join();
    b.join();
    c.join();
    dz_host_pool_set_spin(0);
    run_streams(40, 8, steps / 2, 6);
    // fewer items than workers, one item, and the spin setting changing between parallel-fors
    for (int n : {1, 2, 3, 5}) {
        dz_host_pool_set_spin(n % 2 ? 40 : 0);
        run_streams(50u + n, n, 20, 8);
    }
    printf("host_sanitize ok\n");
    return 0;
}
