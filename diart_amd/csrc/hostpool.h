// Host-side worker pool of the per-stream CPU stages (clustering, aggregation + binarisation).
// The streams of a step are independent, so both stages are a parallel-for over streams; the pool
// keeps its threads between steps (spawning 2 x 8 std::threads per 1.3 ms step cost more than the
// work they did) and hands out stream indices one at a time (streams differ in cost: the number of
// active speakers decides how much of the assignment problem there is).
#pragma once
#include <functional>

// fn(worker, i) for i in [0, n) on at most `threads` threads (the caller is worker 0 and takes part).
// Returns when every index has been processed.  Calls from different host threads are serialised.
void dz_host_parallel(int n, int threads, const std::function<void(int, int)>& fn);
