// Device-resident rolling window of N audio streams.
//
// Replaces, for the N-stream driver, rearrange_audio_stream
// (/root/reference/src/diart/operators.py:44-100: accumulate blocks until `duration` seconds are
// buffered, then emit the last `duration` seconds every `step` seconds) and the per-step upload
// of the whole window that follows it (blocks/segmentation.py:47 / blocks/embedding.py:52: the
// reference moves 320 KB per chunk to the device although only 32 KB of it are new).
//
// Layout: one row of 2*P floats per stream, P = W + slack*hop (W = window samples).  A new
// block of `hop` samples is written at positions p and p + P (p advances modulo P), so the newest
// W samples are ALWAYS one contiguous span [q, q + W) of the row, q = (p + slack*hop) mod P: the
// segmentation / embedding kernels read the rolling window in place through (base pointer +
// offset, row stride 2*P) with their usual coalesced 16-byte loads — no wrap-around logic in any
// kernel, nothing is copied or repeated.  Only the new samples cross PCIe: ONE contiguous
// asynchronous H2D copy into a staging block, then one small kernel writes the block to both
// positions (the 2-D runtime copies this replaced cost the step ~0.3 ms of host and queue time).  The slack keeps a push from overwriting samples a forward
// pass of the previous `slack` windows may still be reading: block t+1 lands on the slot of
// block t+1-W/hop-slack, which belongs to windows <= t-slack only.
//
// Streams that do NOT advance in lock step (StreamServer: every stream has its own pace) use the
// per-row entry points: every row keeps its own write position (dz_ring_push_rows advances only the
// listed rows) and dz_ring_gather copies the current windows of the listed rows, device to device,
// into the dense (k, W) batch the forward passes take — 320 KB per window at HBM speed instead of
// 320 KB per window over PCIe.
#include "dz_common.h"

#include <new>
#include <vector>

struct dz_ring {
    dz_ctx* ctx;
    int n, W, hop, P;
    long long pitch;   // floats between the rows of two streams: 2P rounded up to an ODD multiple of
                       // 256 bytes (2P itself is a large power-of-two multiple for the usual
                       // geometries: every stream's window then starts on the same HBM channel / L2
                       // set and the front-end kernels ran 25 % slower reading the ring than a
                       // dense stream array)
    float* buf;        // [n][2P]
    float* stage[2];   // [n][hop] landing blocks of the H2D copies, used alternately
    long long pushed;  // blocks pushed so far
    int pos;           // write position of the NEXT block, in [0, P)
    // per-row state (dz_ring_push_rows / dz_ring_gather); a lock-step dz_ring_push moves every row
    std::vector<int> rpos;
    std::vector<long long> rpushed;
};

// up to 64 (row, offset) pairs per launch, passed in the kernel-argument segment
struct DzRowList {
    int row[64];
    int off[64];
};

extern "C" int dz_ring_create(dz_ctx* ctx, int n_streams, int window, int hop, int slack_blocks,
                              dz_ring** out) {
    DZ_REQUIRE(ctx && out, "dz_ring_create: NULL argument");
    DZ_REQUIRE(n_streams >= 1 && window >= 4 && hop >= 4 && slack_blocks >= 0,
               "dz_ring_create: empty geometry");
    DZ_REQUIRE(window % hop == 0 && hop % 4 == 0,
               "dz_ring_create: window (%d) must be a multiple of hop (%d) and hop a multiple of 4 "
               "samples (16-byte aligned rows)", window, hop);
    DZ_HIP(hipSetDevice(ctx->device));
    dz_ring* r = new (std::nothrow) dz_ring;
    DZ_REQUIRE(r != nullptr, "dz_ring_create: out of memory");
    r->ctx = ctx; r->n = n_streams; r->W = window; r->hop = hop; r->buf = nullptr;
    r->P = window + slack_blocks * hop;
    r->pushed = 0; r->pos = 0;
    r->rpos.assign(n_streams, 0);
    r->rpushed.assign(n_streams, 0);
    r->pitch = ((2LL * r->P + 63) / 64) * 64;
    if (((r->pitch / 64) & 1) == 0) r->pitch += 64;
    const size_t bytes = (size_t)n_streams * r->pitch * sizeof(float);
    hipError_t e = hipMalloc((void**)&r->buf, bytes);
    if (e != hipSuccess) {
        dz_set_error("dz_ring_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        delete r;
        return 1;
    }
    DZ_HIP(hipMemset(r->buf, 0, bytes));
    r->stage[0] = r->stage[1] = nullptr;
    e = hipMalloc((void**)&r->stage[0], (size_t)2 * n_streams * hop * sizeof(float));
    if (e != hipSuccess) {
        dz_set_error("dz_ring_create: hipMalloc(stage) failed: %s", hipGetErrorString(e));
        (void)hipFree(r->buf);
        delete r;
        return 1;
    }
    r->stage[1] = r->stage[0] + (size_t)n_streams * hop;
    *out = r;
    return 0;
}

// block [n][hop] (rows `bstride` floats apart) -> ring rows at `pos` and `pos + P`; 16 bytes per lane
__global__ __launch_bounds__(256) void ring_scatter_kernel(const float* __restrict__ block, long long bstride,
                                                           float* __restrict__ buf, long long pitch, int n,
                                                           int hop4, int P, int pos) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= hop4) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(block + (long long)i * bstride + 4 * j);
    float* row = buf + (long long)i * pitch + pos + 4 * j;
    *reinterpret_cast<f32x4*>(row) = v;
    *reinterpret_cast<f32x4*>(row + P) = v;
}

// row j of the block -> ring row rows.row[j] at rows.off[j] and rows.off[j] + P
__global__ __launch_bounds__(256) void ring_scatter_rows_kernel(const float* __restrict__ block, long long bstride,
                                                                float* __restrict__ buf, long long pitch,
                                                                int hop4, int P, DzRowList rows) {
    const int jrow = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= hop4) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(block + (long long)jrow * bstride + 4 * j);
    float* row = buf + (long long)rows.row[jrow] * pitch + rows.off[jrow] + 4 * j;
    *reinterpret_cast<f32x4*>(row) = v;
    *reinterpret_cast<f32x4*>(row + P) = v;
}

// out row j <- the W samples of ring row rows.row[j] starting at rows.off[j]
__global__ __launch_bounds__(256) void ring_gather_kernel(const float* __restrict__ buf, long long pitch,
                                                          float* __restrict__ out, long long ostride, int W4,
                                                          DzRowList rows) {
    const int jrow = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= W4) return;
    const float* src = buf + (long long)rows.row[jrow] * pitch + rows.off[jrow] + 4 * j;
    *reinterpret_cast<f32x4*>(out + (long long)jrow * ostride + 4 * j) = *reinterpret_cast<const f32x4*>(src);
}

// ---------------------------------------------------------------------------
// A step's results -> pinned host memory with ONE kernel (stores over the host link; pinned memory is mapped at
// its own address).  Why not hipMemcpyAsync: on this runtime a device-to-host copy call now and then blocks its
// caller until everything queued on that stream has run — measured in `StreamBatch`: about once per 150 steps
// the D2H copy at the end of a step's launch sat 6 - 13 ms in the call (a whole step's latency under load; the
// segmentation copy or the embedding copy, never the kernel launches or the event calls around them), the
// host fed nothing meanwhile and the eight steps in flight drained: -3 to -7 % on a 200-step run, 8 ms of a 19 ms
// driver-form region (profiles/r06z_launch_stalls.json).  A kernel launch does not take that path.
// Replaces the two `.cpu()` of /root/reference/src/diart/blocks/segmentation.py:47 and blocks/embedding.py:68.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void results_to_host_kernel(const float* __restrict__ a, float* __restrict__ ha,
                                                              long long na, const float* __restrict__ b,
                                                              float* __restrict__ hb, long long nb) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // 16-byte piece of (a | b)
    const long long pa = (na + 3) / 4;
    const float* src = a;
    float* dst = ha;
    long long n = na;
    if (i >= pa) { i -= pa; src = b; dst = hb; n = nb; }
    if (4 * i + 3 < n) {
        reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[i];
    } else {
        for (long long j = 4 * i; j < n; ++j) dst[j] = src[j];
    }
}

extern "C" int dz_results_to_host(dz_ctx* ctx, const float* d_a, float* h_a, long long n_a, const float* d_b,
                                  float* h_b, long long n_b, void* stream) {
    DZ_REQUIRE(ctx && d_a && h_a && n_a >= 1, "dz_results_to_host: NULL / empty first buffer");
    DZ_REQUIRE(n_b >= 0 && (n_b == 0 || (d_b && h_b)), "dz_results_to_host: NULL second buffer");
    DZ_REQUIRE((((uintptr_t)d_a | (uintptr_t)h_a | (uintptr_t)d_b | (uintptr_t)h_b) & 15) == 0,
               "dz_results_to_host: buffers must be 16-byte aligned");
    DZ_HIP(hipSetDevice(ctx->device));
    const long long pieces = (n_a + 3) / 4 + (n_b + 3) / 4;
    DZ_LAUNCH(results_to_host_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
              d_a, h_a, n_a, d_b, h_b, n_b);
    DZ_HIP(hipGetLastError());
    return 0;
}

extern "C" int dz_ring_destroy(dz_ring* r) {
    if (r) {
        if (r->buf) (void)hipFree(r->buf);
        if (r->stage[0]) (void)hipFree(r->stage[0]);
        delete r;
    }
    return 0;
}

extern "C" int dz_ring_reset(dz_ring* r) {
    DZ_REQUIRE(r, "dz_ring_reset: NULL argument");
    r->pushed = 0;
    r->pos = 0;
    r->rpos.assign(r->n, 0);
    r->rpushed.assign(r->n, 0);
    return 0;
}

extern "C" int dz_ring_reset_row(dz_ring* r, int row) {
    DZ_REQUIRE(r && row >= 0 && row < r->n, "dz_ring_reset_row: bad row");
    r->rpos[row] = 0;
    r->rpushed[row] = 0;
    return 0;
}

// block: [n][hop] floats with `block_stride` floats between rows; on the host (pinned memory makes
// the copy asynchronous) when on_device == 0, in device memory otherwise.
extern "C" int dz_ring_push(dz_ring* r, const float* block, long long block_stride, int on_device,
                            void* stream) {
    DZ_REQUIRE(r && block, "dz_ring_push: NULL argument");
    DZ_REQUIRE(block_stride >= r->hop, "dz_ring_push: block_stride %lld < hop %d", block_stride,
               r->hop);
    DZ_HIP(hipSetDevice(r->ctx->device));
    hipStream_t st = (hipStream_t)stream;
    DZ_REQUIRE(((uintptr_t)block & 15) == 0 && (block_stride & 3) == 0,
               "dz_ring_push: block rows must be 16-byte aligned");
    const float* src = block;
    long long sstride = block_stride;
    if (on_device == 2) {
        // pinned host memory: ask the runtime for its device-side address and let the scatter kernel
        // read it in place over PCIe (32 KB per stream).  That keeps the upload off the copy queues,
        // where it was seen to wait behind the D2H copy of a step whose results were not ready yet
        // (-17 % with one embedding stream per lane).  Not mappable -> staged copy below.
        void* dptr = nullptr;
        if (hipHostGetDevicePointer(&dptr, (void*)block, 0) == hipSuccess && dptr) {
            src = reinterpret_cast<const float*>(dptr);
        } else {
            (void)hipGetLastError();
            on_device = 0;
        }
    }
    if (!on_device) {
        // pushes of one ring are issued in stream order, so two landing blocks used alternately
        // are never overwritten before the scatter that reads them has run
        float* land = r->stage[r->pushed & 1];
        if (block_stride == r->hop) {
            DZ_HIP(hipMemcpyAsync(land, block, (size_t)r->n * r->hop * sizeof(float),
                                  hipMemcpyHostToDevice, st));
        } else {
            DZ_HIP(hipMemcpy2DAsync(land, (size_t)r->hop * sizeof(float), block,
                                    (size_t)block_stride * sizeof(float), (size_t)r->hop * sizeof(float),
                                    r->n, hipMemcpyHostToDevice, st));
        }
        src = land;
        sstride = r->hop;
    }
    const int hop4 = r->hop / 4;
    hipLaunchKernelGGL(ring_scatter_kernel, dim3((hop4 + 255) / 256, r->n), dim3(256), 0, st, src, sstride,
                       r->buf, r->pitch, r->n, hop4, r->P, r->pos);
    DZ_HIP(hipGetLastError());
    r->pos = (r->pos + r->hop) % r->P;
    r->pushed += 1;
    for (int i = 0; i < r->n; ++i) {
        r->rpos[i] = r->pos;
        r->rpushed[i] += 1;
    }
    return 0;
}

// resolve a host block for the scatter kernels: mapped pinned memory (mode 2) or a staged copy
static int ring_source(dz_ring* r, const float* block, long long block_stride, int rows, int& on_device,
                       const float*& src, long long& sstride, hipStream_t st) {
    src = block;
    sstride = block_stride;
    if (on_device == 2) {
        void* dptr = nullptr;
        if (hipHostGetDevicePointer(&dptr, (void*)block, 0) == hipSuccess && dptr) {
            src = reinterpret_cast<const float*>(dptr);
        } else {
            (void)hipGetLastError();
            on_device = 0;
        }
    }
    if (!on_device) {
        float* land = r->stage[r->pushed & 1];
        DZ_HIP(hipMemcpy2DAsync(land, (size_t)r->hop * sizeof(float), block, (size_t)block_stride * sizeof(float),
                                (size_t)r->hop * sizeof(float), rows, hipMemcpyHostToDevice, st));
        src = land;
        sstride = r->hop;
    }
    return 0;
}

// One new block for each of the `k` listed streams: row j of `block` (k, hop) goes to ring row
// rows[j], at THAT row's own write position.  Rows must be distinct.
extern "C" int dz_ring_push_rows(dz_ring* r, const float* block, long long block_stride, int on_device,
                                 const int* rows, int k, void* stream) {
    DZ_REQUIRE(r && block && rows, "dz_ring_push_rows: NULL argument");
    DZ_REQUIRE(k >= 1 && k <= r->n, "dz_ring_push_rows: %d rows for %d streams", k, r->n);
    DZ_REQUIRE(block_stride >= r->hop && ((uintptr_t)block & 15) == 0 && (block_stride & 3) == 0,
               "dz_ring_push_rows: rows of the block must be >= hop floats apart and 16-byte aligned");
    std::vector<char> seen(r->n, 0);
    for (int j = 0; j < k; ++j) {
        DZ_REQUIRE(rows[j] >= 0 && rows[j] < r->n && !seen[rows[j]], "dz_ring_push_rows: bad / repeated row %d",
                   rows[j]);
        seen[rows[j]] = 1;
    }
    DZ_HIP(hipSetDevice(r->ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const float* src;
    long long sstride;
    int rc = ring_source(r, block, block_stride, k, on_device, src, sstride, st);
    if (rc) return rc;
    const int hop4 = r->hop / 4;
    for (int j0 = 0; j0 < k; j0 += 64) {
        const int m = k - j0 < 64 ? k - j0 : 64;
        DzRowList rl;
        for (int j = 0; j < m; ++j) {
            rl.row[j] = rows[j0 + j];
            rl.off[j] = r->rpos[rows[j0 + j]];
        }
        hipLaunchKernelGGL(ring_scatter_rows_kernel, dim3((hop4 + 255) / 256, m), dim3(256), 0, st,
                           src + (long long)j0 * sstride, sstride, r->buf, r->pitch, hop4, r->P, rl);
        DZ_HIP(hipGetLastError());
    }
    for (int j = 0; j < k; ++j) {
        r->rpos[rows[j]] = (r->rpos[rows[j]] + r->hop) % r->P;
        r->rpushed[rows[j]] += 1;
    }
    r->pushed += 1;        // alternates the landing blocks
    return 0;
}

// samples received by one row so far, capped at the window
extern "C" int dz_ring_filled_row(const dz_ring* r, int row, int* filled) {
    DZ_REQUIRE(r && filled && row >= 0 && row < r->n, "dz_ring_filled_row: bad argument");
    const long long got = r->rpushed[row] * r->hop;
    *filled = got >= r->W ? r->W : (int)got;
    return 0;
}

// d_out row j (out_stride floats apart) <- the current window of ring row rows[j]; every listed row
// must hold a complete window
extern "C" int dz_ring_gather(const dz_ring* r, const int* rows, int k, float* d_out, long long out_stride,
                              void* stream) {
    DZ_REQUIRE(r && rows && d_out, "dz_ring_gather: NULL argument");
    DZ_REQUIRE(k >= 1 && out_stride >= r->W && (out_stride & 3) == 0 && ((uintptr_t)d_out & 15) == 0,
               "dz_ring_gather: output rows must be >= window floats apart and 16-byte aligned");
    for (int j = 0; j < k; ++j) {
        DZ_REQUIRE(rows[j] >= 0 && rows[j] < r->n, "dz_ring_gather: bad row %d", rows[j]);
        DZ_REQUIRE(r->rpushed[rows[j]] * r->hop >= r->W, "dz_ring_gather: the window of row %d is not complete",
                   rows[j]);
    }
    DZ_HIP(hipSetDevice(r->ctx->device));
    const int W4 = r->W / 4;
    for (int j0 = 0; j0 < k; j0 += 64) {
        const int m = k - j0 < 64 ? k - j0 : 64;
        DzRowList rl;
        for (int j = 0; j < m; ++j) {
            rl.row[j] = rows[j0 + j];
            rl.off[j] = (r->rpos[rows[j0 + j]] + r->P - r->W) % r->P;
        }
        hipLaunchKernelGGL(ring_gather_kernel, dim3((W4 + 255) / 256, m), dim3(256), 0, (hipStream_t)stream,
                           r->buf, r->pitch, d_out + (long long)j0 * out_stride, out_stride, W4, rl);
        DZ_HIP(hipGetLastError());
    }
    return 0;
}

// The rolling window as the forward passes take it: *d_wave + i * *stride is the window of stream
// i.  *filled = samples received so far, capped at W: the window is complete (what
// rearrange_audio_stream would emit) once *filled == W.
extern "C" int dz_ring_window(const dz_ring* r, const float** d_wave, long long* stride,
                              int* filled) {
    DZ_REQUIRE(r && d_wave && stride, "dz_ring_window: NULL argument");
    *d_wave = r->buf + (r->pos + r->P - r->W) % r->P;
    *stride = r->pitch;
    if (filled) {
        const long long got = r->pushed * r->hop;
        *filled = got >= r->W ? r->W : (int)got;
    }
    return 0;
}

// contiguous copy (n, W) of the current window (tests; a consumer that wants the window the way
// rearrange_audio_stream emits it).  The forward passes do not need it.
extern "C" int dz_ring_read(const dz_ring* r, float* d_out, void* stream) {
    DZ_REQUIRE(r && d_out, "dz_ring_read: NULL argument");
    DZ_HIP(hipSetDevice(r->ctx->device));
    const float* src = r->buf + (r->pos + r->P - r->W) % r->P;
    DZ_HIP(hipMemcpy2DAsync(d_out, (size_t)r->W * sizeof(float), src,
                            (size_t)r->pitch * sizeof(float), (size_t)r->W * sizeof(float), r->n,
                            hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
