// Host side of the incremental speaker clustering, fp64 like the reference.
//
// Restates /root/reference/src/diart/blocks/clustering.py:119-218 (identify / __call__),
// :85-117 (update / add_center) and the SpeakerMap algebra it uses from
// /root/reference/src/diart/mapping.py (:15-21 optimal_assignments / mapped_indices,
// :217-231 valid_assignments (loose), :245-251 set_source_speaker, :260-294 unmap_threshold /
// unmap_speakers, :341-360 apply).  `lsap` follows scipy.optimize.linear_sum_assignment
// (scipy/optimize/rectangular_lsap: Crouse's shortest augmenting path, columns scanned in
// reverse, ties resolved towards unassigned columns) because the 1e10 sentinels of
// mapping.py:48-52 make ties the normal case and the tie-breaking decides assignments.
#include "../../include/diart_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include "hostpool.h"
#include <utility>
#include <vector>

void dz_set_error(const char* fmt, ...);
extern "C" const char* dz_last_error(void);

namespace {

constexpr double kInvalid = 1e10;  // MinimizationObjective.invalid_value, mapping.py:48-52

// ---------------------------------------------------------------------------
// rectangular LSAP (minimise).  Returns 0 ok, 1 invalid entries, 2 infeasible.
// pairs: (row, col) sorted by row, min(nr, nc) of them.
// ---------------------------------------------------------------------------
int lsap_solve(const double* cost_in, int nr, int nc, std::vector<std::pair<int, int>>& pairs) {
    pairs.clear();
    if (nr == 0 || nc == 0) return 0;
    const bool transpose = nc < nr;
    std::vector<double> temp;
    const double* cost = cost_in;
    if (transpose) {
        temp.resize((size_t)nr * nc);
        for (int i = 0; i < nr; ++i)
            for (int j = 0; j < nc; ++j) temp[(size_t)j * nr + i] = cost_in[(size_t)i * nc + j];
        std::swap(nr, nc);
        cost = temp.data();
    }
    for (size_t i = 0; i < (size_t)nr * nc; ++i)
        if (cost[i] != cost[i] || cost[i] == -INFINITY) return 1;

    std::vector<double> u(nr, 0.0), v(nc, 0.0), spc(nc);
    std::vector<int> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);

    for (int cur = 0; cur < nr; ++cur) {
        // ---- augmenting path from row `cur`
        double minVal = 0.0;
        int num_remaining = nc;
        for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
        std::fill(SR.begin(), SR.end(), 0);
        std::fill(SC.begin(), SC.end(), 0);
        std::fill(spc.begin(), spc.end(), INFINITY);
        int sink = -1, i = cur;
        while (sink == -1) {
            int index = -1;
            double lowest = INFINITY;
            SR[i] = 1;
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = minVal + cost[(size_t)i * nc + j] - u[i] - v[j];
                if (r < spc[j]) {
                    path[j] = i;
                    spc[j] = r;
                }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
                    lowest = spc[j];
                    index = it;
                }
            }
            minVal = lowest;
            if (minVal == INFINITY) return 2;
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j;
            else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        // ---- dual update
        u[cur] += minVal;
        for (int r = 0; r < nr; ++r)
            if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
        for (int j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= minVal - spc[j];
        // ---- augment
        int j = sink;
        while (true) {
            const int r = path[j];
            row4col[j] = r;
            std::swap(col4row[r], j);
            if (r == cur) break;
        }
    }
    if (transpose) {
        // rows of the transposed problem are the original columns
        for (int c = 0; c < nr; ++c) pairs.emplace_back(col4row[c], c);
        std::sort(pairs.begin(), pairs.end());
    } else {
        for (int r = 0; r < nr; ++r) pairs.emplace_back(r, col4row[r]);
    }
    return 0;
}

// ---------------------------------------------------------------------------
// SpeakerMap (minimisation objective only), K x G fp64
// ---------------------------------------------------------------------------
struct SpeakerMap {
    int K = 0, G = 0;
    std::vector<double> m;
    bool solved = false;
    int lsap_rc = 0;
    std::vector<int> raw;  // list(lsap(matrix)[1])  (mapping.py:15-16)

    SpeakerMap(int k, int g) : K(k), G(g), m((size_t)k * g, kInvalid) {}
    double& at(int s, int t) { return m[(size_t)s * G + t]; }
    double at(int s, int t) const { return m[(size_t)s * G + t]; }

    // mapping.py:18-21 + :239-240 — a row is mapped iff its minimum is not the sentinel
    bool source_mapped(int s) const {
        double best = at(s, 0);
        for (int t = 1; t < G; ++t) {
            const double x = at(s, t);
            // np.min propagates NaN
            if (x != x) return true;
            if (x < best) best = x;
        }
        if (best != best) return true;
        return best != kInvalid;
    }
    int solve() {
        if (!solved) {
            std::vector<std::pair<int, int>> pairs;
            lsap_rc = lsap_solve(m.data(), K, G, pairs);
            raw.clear();
            for (auto& p : pairs) raw.push_back(p.second);
            solved = true;
        }
        return lsap_rc;
    }
    // mapping.py:217-231 with strict=False: enumerate(raw) and keep mapped sources
    int valid(std::vector<int>& src, std::vector<int>& tgt) {
        src.clear();
        tgt.clear();
        const int rc = solve();
        if (rc) return rc;
        for (int s = 0; s < (int)raw.size(); ++s)
            if (source_mapped(s)) {
                src.push_back(s);
                tgt.push_back(raw[s]);
            }
        return 0;
    }
    void invalidate() { solved = false; }
    void unmap_source(int s) {
        for (int t = 0; t < G; ++t) at(s, t) = kInvalid;
        invalidate();
    }
    void unmap_target(int t) {
        for (int s = 0; s < K; ++s) at(s, t) = kInvalid;
        invalidate();
    }
    void set_source(int s, int t) {  // mapping.py:245-251, best_possible_value = 0
        at(s, t) = 0.0;
        invalidate();
    }
};

// scipy.spatial.distance.cdist(..., "cosine"), fp64.  The scipy 1.15 x86-64 build sums dot
// products in two interleaved lanes (SSE2 doubles: even / odd elements, lanes added at the
// end, then the odd tail).  The order is reproduced exactly: with duplicated embeddings two
// rows of the cost matrix tie, and which one the Hungarian step favours depends on the last
// bit of the other entries (tests/golden/clustering_crowded.npz step 29 pins this).
#pragma clang fp contract(off)
double dot2(const double* u, const double* v, int n) {
    double s0 = 0.0, s1 = 0.0;
    const int m = n & ~1;
    for (int i = 0; i < m; i += 2) {
        s0 += u[i] * v[i];
        s1 += u[i + 1] * v[i + 1];
    }
    double s = s0 + s1;
    for (int i = m; i < n; ++i) s += u[i] * v[i];
    return s;
}

double cosine_dist(const double* u, const double* v, int n, double nu, double nv) {
    const double dot = dot2(u, v, n);
    double c = dot / (nu * nv);
    if (std::fabs(c) > 1.0) c = std::copysign(1.0, c);
    return 1.0 - c;
}

}  // namespace

struct dz_clu {
    double tau, rho, delta;
    int G;
    int D = 0;
    bool has_centers = false;
    std::vector<double> centers;  // G x D
    std::vector<char> active;     // G
    // blocked_centers is never populated by the reference (clustering.py:46,83)

    int num_known() const {
        int n = 0;
        for (char a : active) n += a ? 1 : 0;
        return n;
    }
    int next_center() const {  // clustering.py:68-71
        for (int c = 0; c < G; ++c)
            if (!active[c]) return c;
        return -1;
    }
    int add_center(const double* e) {  // clustering.py:101-117
        const int c = next_center();
        if (c < 0) return -1;
        std::memcpy(&centers[(size_t)c * D], e, sizeof(double) * D);
        active[c] = 1;
        return c;
    }
};

namespace {

// clustering.py:119-210.  Returns 0 ok; 3 = the reference would raise (assert / scipy error).
int identify(dz_clu* c, const float* seg, int F, int K, const float* emb32, int D, SpeakerMap& out) {
    if (c->has_centers && D != c->D) {
        dz_set_error("clustering: embedding dimension changed from %d to %d", c->D, D);
        return 2;
    }
    std::vector<double> emb((size_t)K * D);
    for (size_t i = 0; i < emb.size(); ++i) emb[i] = (double)emb32[i];

    // :137-145  active: max_f >= tau ; long: mean_f >= rho (float32 arithmetic, like numpy
    // on the float32 segmentation array) ; drop speakers with NaN embeddings
    const float tau32 = (float)c->tau, rho32 = (float)c->rho;
    std::vector<char> is_active(K, 0), is_long(K, 0);
    for (int k = 0; k < K; ++k) {
        float mx = seg[k], sum = 0.f;
        bool nanmax = false;
        for (int f = 0; f < F; ++f) {
            const float x = seg[(size_t)f * K + k];
            if (x != x) nanmax = true;
            if (x > mx) mx = x;
            sum += x;
        }
        const float mean = sum / (float)F;
        bool act = !nanmax && (mx >= tau32);
        is_long[k] = (mean >= rho32) ? 1 : 0;
        bool has_nan = false;
        for (int d = 0; d < D; ++d) {
            const double e = emb[(size_t)k * D + d];
            if (e != e) { has_nan = true; break; }
        }
        is_active[k] = (act && !has_nan) ? 1 : 0;
    }

    if (!c->has_centers) {  // :149-158
        c->D = D;
        c->centers.assign((size_t)c->G * D, 0.0);
        c->active.assign(c->G, 0);
        c->has_centers = true;
        for (int k = 0; k < K; ++k)
            if (is_active[k]) {
                // More active local speakers than max_speakers on the very first chunk (only
                // possible with max_speakers < K): the reference does not raise — its
                // get_next_center_position() returns None and `centers[None] = emb` then overwrites
                // EVERY centroid (clustering.py:101-117), i.e. undefined results.  Here the speakers
                // that found no slot stay unmapped for this chunk (their score columns are zero),
                // like any speaker that cannot be assigned later on.
                const int g = c->add_center(&emb[(size_t)k * D]);
                if (g < 0) continue;
                out.set_source(k, g);
            }
        return 0;
    }

    // :161  cosine distance map (unused centroids are zero vectors -> NaN, overwritten below)
    const int G = c->G;
    SpeakerMap dist(K, G);
    std::vector<double> cn(G);
    for (int g = 0; g < G; ++g) {
        const double* v = &c->centers[(size_t)g * D];
        cn[g] = std::sqrt(dot2(v, v, D));
    }
    for (int k = 0; k < K; ++k) {
        const double* u = &emb[(size_t)k * D];
        const double un = std::sqrt(dot2(u, u, D));
        for (int g = 0; g < G; ++g)
            dist.at(k, g) = cosine_dist(u, &c->centers[(size_t)g * D], D, un, cn[g]);
    }
    // :163-166  invalidate inactive local speakers and inactive centroids
    for (int k = 0; k < K; ++k)
        if (!is_active[k]) dist.unmap_source(k);
    for (int g = 0; g < G; ++g)
        if (!c->active[g]) dist.unmap_target(g);

    // :168  unmap_threshold(delta_new): assignments with dist >= delta are dropped
    SpeakerMap valid = dist;
    {
        std::vector<int> src, tgt;
        if (dist.valid(src, tgt)) {
            dz_set_error("clustering: cost matrix contains invalid numeric entries");
            return 3;
        }
        for (size_t i = 0; i < src.size(); ++i)
            if (dist.at(src[i], tgt[i]) >= c->delta) valid.unmap_source(src[i]);
    }

    // :171-194
    std::vector<int> missed;
    for (int k = 0; k < K; ++k)
        if (is_active[k] && !valid.source_mapped(k)) missed.push_back(k);
    std::vector<int> new_center_speakers;
    const int num_free = G - c->num_known();
    for (int spk : missed) {
        const bool has_space = (int)new_center_speakers.size() < num_free;
        if (has_space && is_long[spk]) {
            new_center_speakers.push_back(spk);
        } else {
            std::vector<int> pref;
            for (int g = 0; g < G; ++g)
                if (c->active[g]) pref.push_back(g);
            std::stable_sort(pref.begin(), pref.end(), [&](int a, int b) {
                return dist.at(spk, a) < dist.at(spk, b);
            });
            std::vector<int> src, tgt;
            if (valid.valid(src, tgt)) {
                dz_set_error("clustering: cost matrix contains invalid numeric entries");
                return 3;
            }
            for (int g : pref)
                if (std::find(tgt.begin(), tgt.end(), g) == tgt.end()) {
                    valid.set_source(spk, g);
                    break;
                }
        }
    }

    // :197-202  update centroids of non-missed long speakers
    {
        std::vector<int> src, tgt;
        if (valid.valid(src, tgt)) {
            dz_set_error("clustering: cost matrix contains invalid numeric entries");
            return 3;
        }
        for (size_t i = 0; i < src.size(); ++i) {
            const int ls = src[i], gs = tgt[i];
            if (std::find(missed.begin(), missed.end(), ls) != missed.end() || !is_long[ls]) continue;
            if (!c->active[gs]) {
                dz_set_error("Cannot update unknown centers");  // clustering.py:98 assert
                return 3;
            }
            double* ctr = &c->centers[(size_t)gs * D];
            const double* e = &emb[(size_t)ls * D];
            for (int d = 0; d < D; ++d) ctr[d] += e[d];
        }
    }
    // :205-208  new centroids
    for (int spk : new_center_speakers) {
        const int g = c->add_center(&emb[(size_t)spk * D]);
        if (g < 0) {
            dz_set_error("clustering: no free center");
            return 3;
        }
        valid.set_source(spk, g);
    }
    out = valid;
    return 0;
}

int step(dz_clu* c, const float* seg, int F, int K, const float* emb, int D, double* scores,
         int* assign) {
    SpeakerMap map(K, c->G);
    const int rc = identify(c, seg, F, K, emb, D, map);
    if (rc) return rc;
    // mapping.py:341-360 apply
    std::vector<int> src, tgt;
    if (map.valid(src, tgt)) {
        dz_set_error("clustering: cost matrix contains invalid numeric entries");
        return 3;
    }
    if (assign)
        for (int k = 0; k < K; ++k) assign[k] = -1;
    if (scores) std::memset(scores, 0, sizeof(double) * (size_t)F * c->G);
    for (size_t i = 0; i < src.size(); ++i) {
        const int s = src[i], t = tgt[i];
        if (assign && s < K) assign[s] = t;
        if (scores)
            for (int f = 0; f < F; ++f) scores[(size_t)f * c->G + t] = (double)seg[(size_t)f * K + s];
    }
    return 0;
}

}  // namespace

extern "C" int dz_clu_create(double tau_active, double rho_update, double delta_new,
                             int max_speakers, dz_clu** out) {
    if (!out || max_speakers < 1) {
        dz_set_error("dz_clu_create: bad arguments");
        return 2;
    }
    dz_clu* c = new (std::nothrow) dz_clu;
    if (!c) {
        dz_set_error("dz_clu_create: out of memory");
        return 1;
    }
    c->tau = tau_active; c->rho = rho_update; c->delta = delta_new; c->G = max_speakers;
    c->active.assign(max_speakers, 0);
    *out = c;
    return 0;
}
extern "C" int dz_clu_reset(dz_clu* c) {
    if (!c) return 2;
    c->has_centers = false;
    c->D = 0;
    c->centers.clear();
    c->active.assign(c->G, 0);
    return 0;
}
extern "C" int dz_clu_destroy(dz_clu* c) {
    delete c;
    return 0;
}
extern "C" int dz_clu_step(dz_clu* c, const float* seg, int frames, int k_local, const float* emb,
                           int dim, double* scores_out, int* assign_out) {
    if (!c || !seg || !emb || frames < 1 || k_local < 1 || dim < 1) {
        dz_set_error("dz_clu_step: bad arguments");
        return 2;
    }
    return step(c, seg, frames, k_local, emb, dim, scores_out, assign_out);
}
extern "C" int dz_clu_step_batch(dz_clu** clus, int n, const float* seg, int frames, int k_local,
                                 const float* emb, int dim, double* scores_out, int* assign_out,
                                 int num_threads) {
    if (!clus || n < 1 || !seg || !emb || frames < 1 || k_local < 1 || dim < 1) {
        dz_set_error("dz_clu_step_batch: bad arguments");
        return 2;
    }
    const int G = clus[0]->G;
    for (int i = 0; i < n; ++i)
        if (!clus[i] || clus[i]->G != G) {
            dz_set_error("dz_clu_step_batch: handles must share max_speakers");
            return 2;
        }
    auto run = [&](int i) -> int {
        return step(clus[i], seg + (size_t)i * frames * k_local, frames, k_local,
                    emb + (size_t)i * k_local * dim, dim,
                    scores_out ? scores_out + (size_t)i * frames * G : nullptr,
                    assign_out ? assign_out + (size_t)i * k_local : nullptr);
    };
    int nt = num_threads < 1 ? 1 : num_threads;
    if (nt > n) nt = n;
    if (nt == 1) {
        for (int i = 0; i < n; ++i) {
            const int rc = run(i);
            if (rc) return rc;
        }
        return 0;
    }
    // the error text is thread local: a failing worker copies its own message (and which stream
    // it was) before another stream's overwrites it.  Streams are independent, so the ones that
    // succeeded HAVE been stepped; the caller decides what to do with the failed ones.
    std::vector<int> rcs(nt, 0), who(nt, -1);
    std::vector<std::string> msgs(nt);
    dz_host_parallel(n, nt, [&](int t, int i) {
        const int rc = run(i);
        if (rc && (!rcs[t] || i < who[t])) {
            rcs[t] = rc;
            who[t] = i;
            msgs[t] = dz_last_error();
        }
    });
    int first = -1;
    for (int t = 0; t < nt; ++t)
        if (rcs[t] && (first < 0 || who[t] < who[first])) first = t;
    if (first >= 0) {
        int failed = 0;
        for (int rc : rcs) failed += rc != 0;
        dz_set_error("dz_clu_step_batch: stream %d of %d failed (code %d): %s%s", who[first], n, rcs[first],
                     msgs[first].c_str(), failed > 1 ? " (other streams failed too)" : "");
        return rcs[first];
    }
    return 0;
}
extern "C" int dz_clu_get_centers(dz_clu* c, double* out, int dim) {
    if (!c) return 2;
    if (!c->has_centers) return 1;
    if (dim != c->D || !out) {
        dz_set_error("dz_clu_get_centers: dim %d != %d", dim, c->D);
        return 2;
    }
    std::memcpy(out, c->centers.data(), sizeof(double) * c->centers.size());
    return 0;
}
extern "C" int dz_clu_get_active(dz_clu* c, int* out_mask) {
    if (!c || !out_mask) return 2;
    for (int g = 0; g < c->G; ++g) out_mask[g] = c->active[g] ? 1 : 0;
    return 0;
}
extern "C" int dz_clu_dim(dz_clu* c) { return (c && c->has_centers) ? c->D : 0; }
extern "C" int dz_clu_set_state(dz_clu* c, const double* centers, const int* active_mask, int dim) {
    if (!c || !centers || !active_mask || dim < 1) {
        dz_set_error("dz_clu_set_state: bad arguments");
        return 2;
    }
    c->D = dim;
    c->centers.assign(centers, centers + (size_t)c->G * dim);
    for (int g = 0; g < c->G; ++g) c->active[g] = active_mask[g] ? 1 : 0;
    c->has_centers = true;
    return 0;
}
extern "C" int dz_lsap(const double* cost, int nr, int nc, int* col4row) {
    std::vector<std::pair<int, int>> pairs;
    const int rc = lsap_solve(cost, nr, nc, pairs);
    if (rc) {
        dz_set_error(rc == 1 ? "matrix contains invalid numeric entries" : "cost matrix is infeasible");
        return rc;
    }
    for (int r = 0; r < nr; ++r) col4row[r] = -1;
    for (auto& p : pairs) col4row[p.first] = p.second;
    return 0;
}
