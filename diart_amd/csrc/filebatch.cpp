// File-parallel evaluation (BASELINE.json configs 1 / 4): the host half of a GPU step whose rows
// are the next few CONSECUTIVE windows of several files of one rank.
//
// The reference's Benchmark feeds one file at a time, 32 consecutive windows per batch, through
// SpeakerDiarization.__call__ (/root/reference/src/diart/inference.py:392-432,
// blocks/diarization.py:193-232): segmentation / embedding are batched, then a Python loop runs
// clustering -> DelayedAggregation -> Binarize chunk by chunk.  Only that loop is sequential, and
// only WITHIN a file.  Here the rows of one GPU batch are file-major: file i contributes
// count[i] consecutive windows starting at row row0[i]; a host thread takes a whole file and walks
// its windows in order (dz_clu_step -> dz_tail_step), files run in parallel.  No barrier per window.
#include <cstddef>
#include <string>
#include <vector>

#include "../../include/diart_amd.h"
#include "hostpool.h"

void dz_set_error(const char* fmt, ...);

extern "C" int dz_file_step_batch(dz_clu** clus, dz_tail** tails, int n_files, const int* row0,
                                  const int* count, const float* seg, int frames, int k_local,
                                  const float* emb, int dim, int max_speakers, const double* chunk_start,
                                  double resolution, double* turns_out, int max_turns, int* nturns_out,
                                  int* assign_out, int num_threads) {
    if (!clus || !tails || n_files < 1 || !row0 || !count || !seg || !emb || !chunk_start || !turns_out ||
        !nturns_out) {
        dz_set_error("dz_file_step_batch: NULL argument");
        return 2;
    }
    if (frames < 1 || k_local < 1 || dim < 1 || max_speakers < 1 || max_turns < 1) {
        dz_set_error("dz_file_step_batch: empty shape");
        return 2;
    }
    for (int i = 0; i < n_files; ++i)
        if (!clus[i] || !tails[i] || count[i] < 0 || row0[i] < 0 || dz_tail_max_rows(tails[i]) != frames + 2) {
            dz_set_error("dz_file_step_batch: bad handle / row range of file %d", i);
            return 2;
        }
    int nt = num_threads < 1 ? 1 : num_threads;
    if (nt > n_files) nt = n_files;
    const size_t F = (size_t)frames, G = (size_t)max_speakers;
    // per worker: the permuted scores of the current window and the aggregated region (discarded:
    // the RTTM needs the speech turns only)
    std::vector<std::vector<double>> scores(nt, std::vector<double>(F * G)), agg(nt, std::vector<double>((F + 2) * G));
    // the error text is thread local: a failing worker keeps its own message and which file it was
    std::vector<int> rcs(nt, 0), who(nt, -1);
    std::vector<std::string> msgs(nt);
    auto run = [&](int worker, int i) {
        int rows = 0;
        double t0 = 0.0, res = 0.0;
        std::vector<int> assign((size_t)k_local);
        for (int t = 0; t < count[i] && !rcs[worker]; ++t) {
            const size_t r = (size_t)row0[i] + t;
            int rc = dz_clu_step(clus[i], seg + r * F * k_local, frames, k_local, emb + r * (size_t)k_local * dim, dim,
                                 scores[worker].data(), assign_out ? assign_out + r * k_local : assign.data());
            if (!rc)
                rc = dz_tail_step(tails[i], scores[worker].data(), chunk_start[r], resolution, agg[worker].data(), &rows,
                                  &t0, &res, turns_out + r * (size_t)max_turns * 3, max_turns, nturns_out + r);
            if (rc) {
                rcs[worker] = rc;
                who[worker] = i;
                msgs[worker] = dz_last_error();
            }
        }
    };
    if (nt == 1) {
        for (int i = 0; i < n_files && !rcs[0]; ++i) run(0, i);
    } else {
        dz_host_parallel(n_files, nt, run);
    }
    for (int t = 0; t < nt; ++t)
        if (rcs[t]) {
            dz_set_error("dz_file_step_batch: file %d of %d failed (code %d): %s", who[t], n_files, rcs[t],
                         msgs[t].c_str());
            return rcs[t];
        }
    return 0;
}
