// Per-stream output tail of the diarization step, on the host in fp64:
//
//   DelayedAggregation   /root/reference/src/diart/blocks/aggregation.py:120-218
//     strategies         :73-118  (hamming :95-118, mean :73-92 ("average"), first :60-70)
//     first-chunk prepend :188-211
//   Binarize             /root/reference/src/diart/blocks/utils.py:11-59
//   the buffer handling of SpeakerDiarization.__call__
//                        /root/reference/src/diart/blocks/diarization.py:203-232
//
// The reference wraps every buffered prediction and a fresh Hamming window in
// pyannote.core.SlidingWindowFeature objects, crops each of them and walks the frames in a Python
// loop to build the Annotation: per chunk that is ~100 us of interpreter time, i.e. more than the
// GPU spends on the chunk.  Here the frame range of the output region is computed once per
// buffer from the frame grid with the same fp64 expressions pyannote.core evaluates
// (SlidingWindow.crop with return_ranges / fixed duration), rows outside a buffer repeat its
// first / last row, and the sums run in the buffer order numpy's axis-0 reduction uses, so the
// aggregated scores are bit-identical to the reference's and the speech turns come out of one
// pass over rising / falling edges of `score > threshold`.
#include <math.h>
#include <stddef.h>
#include <string.h>

#include <deque>
#include <new>
#include "hostpool.h"
#include <vector>

#include "../../include/diart_amd.h"

void dz_set_error(const char* fmt, ...);

namespace {

struct Buffer {
    std::vector<double> data;  // [F][G]
    double start, res;         // frame grid: frame i covers [start + i*res, start + (i+1)*res)
};

inline double seg_duration(double s, double e) { return e > s ? e - s : 0.0; }

// pyannote.core SlidingWindow.samples(from_duration, mode)
inline long samples_for(double from_duration, double dur, double step, int mode) {
    if (mode == DZ_CROP_STRICT) return (long)floor((from_duration - dur) / step) + 1;
    if (mode == DZ_CROP_LOOSE) return (long)floor((from_duration + dur) / step);
    return (long)nearbyint(from_duration / step);  // center; np.rint = round half to even
}
inline long closest_frame(double t, double start, double dur, double step) {
    return (long)nearbyint((t - start - 0.5 * dur) / step);
}
// first frame index and frame count of focus [fs, fe) on the grid (start, res) for a crop with
// fixed = focus duration
inline void crop_range(double fs, double fe, double start, double res, int mode, long* first,
                       long* count) {
    long i;
    if (mode == DZ_CROP_LOOSE)
        i = (long)ceil((fs - res - start) / res);
    else if (mode == DZ_CROP_STRICT)
        i = (long)ceil((fs - start) / res);
    else
        i = closest_frame(fs, start, res, res);
    *first = i;
    *count = samples_for(seg_duration(fs, fe), res, res, mode);
}
inline int clip(long v, int hi) { return v < 0 ? 0 : (v > hi ? hi : (int)v); }

}  // namespace

struct dz_tail {
    int F, G, strategy, mode, nwin;
    double step, latency, threshold;
    std::vector<double> hamming;  // [F]
    std::deque<Buffer> buffers;
    std::vector<double> num, den;  // scratch
};

namespace {

int tail_step(dz_tail* t, const double* scores, double chunk_start, double res, double* agg_out,
              int* rows_out, double* t0_out, double* res_out, double* turns_out, int max_turns,
              int* nturns_out) {
    const int F = t->F, G = t->G;
    if (!(res > 0.0)) return 2;
    t->buffers.emplace_back();
    Buffer& nb = t->buffers.back();
    nb.data.assign(scores, scores + (size_t)F * G);
    nb.start = chunk_start;
    nb.res = res;
    // buffers[-1].extent.end - latency  (aggregation.py:214-216); extent = start + (n-1)*step + duration
    const double ext_end = nb.start + (F - 1) * nb.res + nb.res;
    const double rs = ext_end - t->latency;
    const double re = rs + t->step;
    const int nbuf = (int)t->buffers.size();
    const int max_rows = F + 2;

    long first0, count;
    crop_range(rs, re, t->buffers[0].start, t->buffers[0].res, t->mode, &first0, &count);
    if (count < 1 || count > max_rows) return 4;
    int rows = (int)count;
    // ---- aggregate the region over the buffers ---------------------------------------
    if (t->strategy == DZ_AGG_FIRST) {
        const Buffer& b = t->buffers[0];
        for (int r = 0; r < rows; ++r)
            memcpy(agg_out + (size_t)r * G, b.data.data() + (size_t)clip(first0 + r, F - 1) * G,
                   sizeof(double) * G);
    } else {
        t->num.assign((size_t)rows * G, 0.0);
        t->den.assign(rows, 0.0);
        for (int bi = 0; bi < nbuf; ++bi) {
            const Buffer& b = t->buffers[bi];
            long first, cnt;
            crop_range(rs, re, b.start, b.res, t->mode, &first, &cnt);
            if (cnt != count) return 4;  // np.stack would raise in the reference
            for (int r = 0; r < rows; ++r) {
                const int row = clip(first + r, F - 1);
                const double* src = b.data.data() + (size_t)row * G;
                double* dst = t->num.data() + (size_t)r * G;
                if (t->strategy == DZ_AGG_HAMMING) {
                    const double h = t->hamming[row];
                    if (bi == 0) {
                        for (int g = 0; g < G; ++g) dst[g] = h * src[g];
                        t->den[r] = h;
                    } else {
                        for (int g = 0; g < G; ++g) dst[g] += h * src[g];
                        t->den[r] += h;
                    }
                } else {
                    if (bi == 0)
                        for (int g = 0; g < G; ++g) dst[g] = src[g];
                    else
                        for (int g = 0; g < G; ++g) dst[g] += src[g];
                }
            }
        }
        for (int r = 0; r < rows; ++r) {
            const double d = t->strategy == DZ_AGG_HAMMING ? t->den[r] : (double)nbuf;
            for (int g = 0; g < G; ++g) agg_out[(size_t)r * G + g] = t->num[(size_t)r * G + g] / d;
        }
    }
    double out_start = rs;
    double out_res = seg_duration(rs, re) / rows;
    // ---- first buffer of a stream: everything up to the end of the region (aggregation.py:188-211)
    if (nbuf == 1 && t->buffers[0].start == 0.0) {
        const Buffer& b = t->buffers[0];
        long f1, c1;
        crop_range(0.0, re, b.start, b.res, t->mode, &f1, &c1);
        if (c1 < rows || c1 > max_rows) return 4;
        const int all = (int)c1;
        // the aggregated rows become the LAST `rows` rows; move them first (ranges may overlap)
        memmove(agg_out + (size_t)(all - rows) * G, agg_out, sizeof(double) * (size_t)rows * G);
        for (int r = 0; r < all - rows; ++r)
            memcpy(agg_out + (size_t)r * G, b.data.data() + (size_t)clip(f1 + r, F - 1) * G,
                   sizeof(double) * G);
        rows = all;
        out_start = 0.0;
        out_res = re / rows;
    }
    *rows_out = rows;
    *t0_out = out_start;
    *res_out = out_res;
    // ---- Binarize (utils.py:43-59): strict `>`; a turn spans [middle(first active frame),
    // middle(first inactive frame after it)), the frame after the last one closing open turns
    int nt = 0;
    if (turns_out) {
        auto middle = [&](int i) {
            const double s = out_start + i * out_res;
            return 0.5 * (s + (s + out_res));
        };
        for (int g = 0; g < G; ++g) {
            int onset = -1;
            for (int r = 0; r <= rows; ++r) {
                const bool on = r < rows && agg_out[(size_t)r * G + g] > t->threshold;
                if (on && onset < 0) onset = r;
                if (!on && onset >= 0) {
                    if (nt >= max_turns) return 5;
                    turns_out[3 * nt + 0] = middle(onset);
                    turns_out[3 * nt + 1] = middle(r);
                    turns_out[3 * nt + 2] = (double)g;
                    ++nt;
                    onset = -1;
                }
            }
        }
    }
    if (nturns_out) *nturns_out = nt;
    // diarization.py:228-232
    if ((int)t->buffers.size() == t->nwin) t->buffers.pop_front();
    return 0;
}

}  // namespace

extern "C" int dz_tail_create(int frames, int speakers, double step, double latency,
                              double threshold, int strategy, int cropping_mode,
                              const double* hamming, dz_tail** out) {
    if (!out || frames < 1 || speakers < 1 || !(step > 0.0) || !(latency >= step) ||
        strategy < DZ_AGG_HAMMING || strategy > DZ_AGG_FIRST || cropping_mode < DZ_CROP_STRICT ||
        cropping_mode > DZ_CROP_CENTER || (strategy == DZ_AGG_HAMMING && !hamming)) {
        dz_set_error("dz_tail_create: bad arguments (latency must be >= step)");
        return 2;
    }
    dz_tail* t = new (std::nothrow) dz_tail;
    if (!t) {
        dz_set_error("dz_tail_create: out of memory");
        return 1;
    }
    t->F = frames; t->G = speakers; t->strategy = strategy; t->mode = cropping_mode;
    t->step = step; t->latency = latency; t->threshold = threshold;
    t->nwin = (int)nearbyint(latency / step);  // int(round(latency / step)), aggregation.py:157
    if (hamming) t->hamming.assign(hamming, hamming + frames);
    *out = t;
    return 0;
}
extern "C" int dz_tail_reset(dz_tail* t) {
    if (!t) return 2;
    t->buffers.clear();
    return 0;
}
extern "C" int dz_tail_destroy(dz_tail* t) {
    delete t;
    return 0;
}
extern "C" int dz_tail_max_rows(const dz_tail* t) { return t ? t->F + 2 : 0; }

static int tail_fail(const char* who, int rc) {
    if (rc == 4)
        dz_set_error("%s: the output region does not map onto the frame grid of every buffer", who);
    else if (rc == 5)
        dz_set_error("%s: more speech turns than max_turns", who);
    else if (rc)
        dz_set_error("%s: bad arguments", who);
    return rc;
}

extern "C" int dz_tail_step(dz_tail* t, const double* scores, double chunk_start, double resolution,
                            double* agg_out, int* rows_out, double* t0_out, double* res_out,
                            double* turns_out, int max_turns, int* nturns_out) {
    if (!t || !scores || !agg_out || !rows_out || !t0_out || !res_out) {
        dz_set_error("dz_tail_step: NULL argument");
        return 2;
    }
    return tail_fail("dz_tail_step", tail_step(t, scores, chunk_start, resolution, agg_out, rows_out,
                                               t0_out, res_out, turns_out, max_turns, nturns_out));
}

extern "C" int dz_tail_step_batch(dz_tail** tails, int n, const double* scores,
                                  const double* chunk_start, const double* resolution,
                                  double* agg_out, int* rows_out, double* t0_out, double* res_out,
                                  double* turns_out, int max_turns, int* nturns_out,
                                  int num_threads) {
    if (!tails || n < 1 || !scores || !chunk_start || !resolution || !agg_out || !rows_out ||
        !t0_out || !res_out) {
        dz_set_error("dz_tail_step_batch: NULL argument");
        return 2;
    }
    const int F = tails[0]->F, G = tails[0]->G;
    for (int i = 0; i < n; ++i)
        if (!tails[i] || tails[i]->F != F || tails[i]->G != G) {
            dz_set_error("dz_tail_step_batch: handles must share frames / speakers");
            return 2;
        }
    const size_t mr = (size_t)F + 2;
    auto run = [&](int i) -> int {
        return tail_step(tails[i], scores + (size_t)i * F * G, chunk_start[i], resolution[i],
                         agg_out + (size_t)i * mr * G, rows_out + i, t0_out + i, res_out + i,
                         turns_out ? turns_out + (size_t)i * max_turns * 3 : nullptr, max_turns,
                         nturns_out ? nturns_out + i : nullptr);
    };
    int nt = num_threads < 1 ? 1 : num_threads;
    if (nt > n) nt = n;
    std::vector<int> rcs(nt, 0);
    if (nt == 1) {
        for (int i = 0; i < n && !rcs[0]; ++i) rcs[0] = run(i);
    } else {
        dz_host_parallel(n, nt, [&](int k, int i) {
            const int rc = run(i);
            if (rc && !rcs[k]) rcs[k] = rc;
        });
    }
    for (int rc : rcs)
        if (rc) return tail_fail("dz_tail_step_batch", rc);
    return 0;
}
