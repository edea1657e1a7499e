#!/usr/bin/env python
"""bench.py — real-time streams per GPU of diart's per-chunk hot path on MI355X.

A "step" = one pass of the hot path over one batch of synthetic input: the current 5 s window of
each of the 64 concurrent 16 kHz streams of one GPU (BASELINE.json configs[1]) goes through
SpeakerSegmentation -> OverlappedSpeechPenalty -> SpeakerEmbedding (x3 speakers) -> normalisation
-> OnlineSpeakerClustering -> DelayedAggregation -> Binarize: lines 186-232 of the reference's
SpeakerDiarization.__call__ for every stream (the audio passthrough of :205 aside — the audio
stays in HBM), down to the speech turns of the 500 ms region each step finalises.
Windows advance by 500 ms per step and are read in place from the streams, which are resident in
HBM before the timed region starts.

    value = chunks/s / 2  = number of real-time streams the job sustains at a 500 ms step
            (each live stream emits 2 chunks per second), whole job, all GPUs.

N > 1 (launched by torch.distributed.run): every rank owns 64 streams of its own (weak scaling,
streams are independent — no data-path collective); rank 0 synthesises the weights and broadcasts
them over RCCL; timing is barrier + synchronize on both sides, max over ranks.

Besides the contract's fields the JSON line carries: ``roofline`` (dominant MFMA-bound kernel) and
``roofline_kernels`` (every kernel against its own bound) from per-kernel dispatch timestamps taken
on every 10th timed step; ``exact_f32`` — the same job on the exact-f32 matrix path (the default
arithmetic is "f16x3"); ``host_fed`` — the same job with each step's new audio
uploaded from pinned host memory into the device ring (PCIe-inclusive); ``cpu_baseline`` — the
oracle on the host cores (rank 0, N = 1).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the HIP runtime starts (diart_amd/__init__.py)
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

F_SEG, F_E1, F_E2, F_E3 = 293, 289, 285, 279
# Per launch tag: algorithmic MACs per 5 s chunk, algorithmic HBM bytes per chunk
# (the layer's input + output read / written once, f32) and weight bytes per LAUNCH.
KERNELS = {
    "wave_stats": dict(mac=0, io=320_000, w=0, bound="hbm"),
    # y2 [293][64] f32 -> InstanceNorm + LeakyReLU -> two f16 planes (k_gemm_split.hip norm_split_kernel), once per network
    "norm_split": dict(mac=0, io=F_SEG * 64 * (4 + 2 * 2), w=0, bound="hbm"),
    # tile partials -> InstanceNorm scale / shift (exact-f32 path and DZ_FUSED_NORM=0 only: launch-bound)
    "finalize_norm": dict(mac=0, io=(10 * 64 * 2 + 2 * 64) * 4, w=0, bound="hbm"),
    "sinc_conv0": dict(mac=160_138_000, io=320_000 + 2658 * 80 * 4, w=128 * 96 * 4, bound="mfma_f32"),
    # the first stage of BOTH networks in one launch (default precision; k_front.hip sinc_conv0_pair_kernel)
    "sinc_conv0_pair": dict(mac=2 * 160_138_000, io=320_000 + 2 * 2658 * 80 * 4, w=2 * 192 * 256 * 2, bound="gemm"),
    "conv1_pool": dict(mac=63_696_000, io=(2658 * 80 + 884 * 64) * 4, w=64 * 416 * 4, bound="gemm"),
    "conv2_pool": dict(mac=15_840_000, io=(884 * 64 + 293 * 64) * 4, w=64 * 320 * 4, bound="gemm"),
    "lstm_proj0": dict(mac=F_SEG * 60 * 1024, io=F_SEG * (64 + 1024) * 4, w=1024 * 64 * 4, bound="gemm"),
    "lstm_proj": dict(mac=F_SEG * 256 * 1024, io=F_SEG * (256 + 1024) * 4, w=1024 * 256 * 4, bound="gemm"),
    "lstm_rec": dict(mac=2 * F_SEG * 512 * 128, io=F_SEG * (1024 + 256) * 4, w=2 * 512 * 128 * 4, bound="rec"),
    # default precision: linear[0] -> linear[1] -> classifier -> activation -> OSP weights, ONE launch
    # (k_mlp_head.hip); with DZ_MLP_HEAD=0 / exact f32 the tag holds the two MLP GEMMs (half of this each)
    "seg_mlp": dict(mac=F_SEG * (256 * 128 + 128 * 128 + 128 * 3), io=F_SEG * (256 * 4 + (3 + 3) * 4),
                    w=(128 * 256 + 128 * 128) * 4 + 3 * 128 * 4, bound="gemm"),
    # Linear(128->3) + sigmoid + OverlappedSpeechPenalty weights in one launch (k_pool.hip seg_head_kernel)
    "seg_classifier": dict(mac=F_SEG * 128 * 3, io=F_SEG * (128 + 3 + 3) * 4, w=3 * 128 * 4, bound="hbm"),
    "seg_head": dict(mac=F_SEG * (256 * 128 + 128 * 128 + 128 * 3), io=F_SEG * (256 + 3 + 3) * 4,
                     w=(128 * 256 + 128 * 128 + 8 * 128) * 4, bound="gemm"),
    "tdnn1": dict(mac=F_E1 * 512 * 300, io=F_SEG * (64 + 512) * 4, w=512 * 320 * 4, bound="gemm"),
    "tdnn2": dict(mac=F_E2 * 512 * 1536, io=F_SEG * 1024 * 4, w=512 * 1536 * 4, bound="gemm"),
    "tdnn3": dict(mac=F_E3 * 512 * 1536, io=F_SEG * 1024 * 4, w=512 * 1536 * 4, bound="gemm"),
    "tdnn4": dict(mac=F_E3 * 512 * 512, io=F_SEG * 1024 * 4, w=512 * 512 * 4, bound="gemm"),
    "tdnn5": dict(mac=F_E3 * 1500 * 512, io=F_SEG * (512 + 1536) * 4, w=1536 * 512 * 4, bound="gemm"),
    "stats_pool": dict(mac=0, io=F_E3 * 1536 * 4 + 3 * F_SEG * 4 + 3 * 3008 * 4, w=0, bound="hbm"),
    "emb_linear": dict(mac=3 * 3000 * 512, io=3 * (3008 + 512) * 4, w=512 * 3008 * 4, bound="mfma_f32"),
}
POOL_PIECES = 4    # 128-row tiles a 293-row chunk can touch (dz_pool_pieces)


def kernels_for(precision):
    """KERNELS as the launches of this precision really are: on the default path tdnn5 keeps its output
    tile in LDS and writes per-tile weighted moments (k_gemm_pre.hip pooled epilogue), and the `stats_pool`
    tag is the small kernel that merges them (pool_combine)."""
    k = {n: dict(v) for n, v in KERNELS.items()}
    if not EXPERIMENTS:
        k.pop("sinc_conv0_pair", None)               # (the pair launch exists in the experiments build only)
    if precision != "f16x3":
        if os.environ.get("DZ_F32_GEMM", "1") == "0":
            k.pop("norm_split", None)                # (round-1 exact-f32 kernels everywhere: the consumers normalise on load)
        else:                                        # exact f32 (round 6): the same pass with f32 rows out (norm_f32_kernel)
            k["norm_split"]["io"] = F_SEG * 64 * (4 + 4)
    if precision == "f16x3" and os.environ.get("DZ_POOL_FUSE", "1") != "0":
        moments = POOL_PIECES * 3 * 1536 * 2 * 4
        k["tdnn5"]["io"] = F_SEG * 512 * 4 + moments + 3 * F_SEG * 4
        k["stats_pool"]["io"] = moments + 3 * 3008 * 4
    return k


# sinc_conv0 folds the (anti)symmetric FIR bank: 42 tiles x 192 frames x 96 columns x 128 taps
EXECUTED_MAC = {"sinc_conv0": 42 * 192 * 96 * 128, "sinc_conv0_pair": 84 * 96 * 192 * 256}
ALG_GFLOP_PER_CHUNK = 3.352   # 1.312 seg + 2.039 emb de-duplicated
# Peaks from /opt/skills/guides/MI355X_MICROARCH.md (chip level, 256 CUs @ 2.4 GHz):
PEAK_F16_MATRIX_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_f16 / 16x16x32_f16, dense
SPLIT_PRODUCTS = 3               # f16 MFMAs per algorithmic product on the split path
PEAK_F32_MATRIX_TFLOPS = 157.3   # v_mfma_f32_16x16x4_f32, dense
PEAK_F32_VECTOR_TFLOPS = 157.3   # f32 vector peak (packed FMA); plain v_fma_f32 issues at half of it
PEAK_HBM_GBPS = 8000.0


STREAMS_PER_GPU = [64]
# Kernel-selection switches are honoured by the experiments build only (DZ_EXPERIMENTS=1, diart_amd/_lib.py);
# the shipped library has one configuration per layer, and this file names ITS kernels by default.
EXPERIMENTS = os.environ.get("DZ_EXPERIMENTS", "0") not in ("", "0")


def xenv(name, default=""):
    return os.environ.get(name, default) if EXPERIMENTS else default



RECURRENCE = {}      # precision -> "valu" | "0" | "3", filled when the engines are built


def lstm_chains_per_wg():
    """k_lstm.hip: one chain per workgroup unless DZ_LSTM_NC=2 (experiment)."""
    e = xenv("DZ_LSTM_NC", "")
    return 2 if e == "2" else 1


def gemm_generation():
    """k_gemm_pre.hip dispatches its launches to k_gemm_g2.hip with DZ_GEMM_GEN=2 (dz_gemm_gen())."""
    g = xenv("DZ_GEMM_GEN", "1")
    return int(g) if g in ("2", "3") else 1


def device_kernel(tag, precision):
    """bench tag -> (rocprofv3 kernel symbol, bound, chip peak, unit): the roofline is reported per
    DEVICE kernel, so the layers that share one instantiation are one entry."""
    split = precision == "f16x3"
    pre = split                                                       # wide layers on k_gemm_pre.hip
    # the recurrence the ENGINE of this precision runs (StreamBatch.recurrence: the matrix-core form for >= 64 streams
    # unless --recurrence says otherwise), else the model's own ("valu")
    lstm = RECURRENCE.get(precision) or "valu"
    k = KERNELS[tag]
    fused_pool = split and pre and os.environ.get("DZ_POOL_FUSE", "1") != "0"
    if k["bound"] == "hbm":
        return {"wave_stats": "wave_stats_kernel", "finalize_norm": "finalize_norm_kernel",
                "norm_split": "norm_split_kernel" if split else "norm_f32_kernel",
                "stats_pool": "pool_combine_kernel" if fused_pool else "stats_pool_reg_kernel<3, 72>",
                "seg_classifier": "seg_head_kernel"}[tag], "hbm", PEAK_HBM_GBPS, "GB/s"
    if k["bound"] == "rec":
        if split and lstm != "valu":
            sym = {"0": "lstm_mfma_kernel<true>", "3": "lstm_mfma_dma_kernel", "4": "lstm_mfma_pipe_kernel<false, 0>"}.get(
                lstm, "lstm_mfma1_kernel<true, %d>" % (0 if lstm == "1" else 8))
            return sym, "mfma", PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS, "TFLOP/s"
        pk = "false" if xenv("DZ_LSTM_PK", "1") == "0" else "true"      # packed-FMA template argument
        return f"lstm_rec_kernel<true, {lstm_chains_per_wg()}, {pk}>", "valu", PEAK_F32_VECTOR_TFLOPS, "TFLOP/s"
    if tag == "sinc_conv0_pair":
        return "sinc_conv0_pair_kernel", "mfma", PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS, "TFLOP/s"
    if tag == "sinc_conv0" and split and xenv("DZ_CONV0_SPLIT", "1") != "0":
        sym = "sinc_conv0_h_kernel<0>" if xenv("DZ_CONV0_V2", "1") == "0" else "sinc_conv0_v2_kernel"
        return sym, "mfma", PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS, "TFLOP/s"
    if k["bound"] == "mfma_f32" or not split:
        sym = {"sinc_conv0": "sinc_conv0_kernel", "conv1_pool": "convgemm_kernel<64, true, 4>",
               "conv2_pool": "convgemm_kernel<64, true, 4>", "lstm_proj": "convgemm_kernel<128, false, 0>",
               "lstm_proj0": "convgemm_kernel<128, true, 0>", "seg_mlp": "convgemm_kernel<128, false, 1>",
               "tdnn1": "convgemm_kernel<128, true, 3>",
               "emb_linear": "convgemm_kernel<128, false, 0> (split-K)", "seg_head": "seg_head_kernel"}
        if os.environ.get("DZ_F32_GEMM", "1") != "0":                  # wide layers without a prologue: k_gemm_f32.hip
            # (round 6: y2 is normalised once, norm_f32_kernel, and the first projection / tdnn1 are such layers too)
            sym.update({"lstm_proj0": "gemm_f32_kernel<0>", "tdnn1": "gemm_f32_kernel<3>",
                        "lstm_proj": "gemm_f32_kernel<0>", "seg_mlp": "gemm_f32_kernel<1>", "tdnn2": "gemm_f32_kernel<3>",
                        "tdnn3": "gemm_f32_kernel<3>", "tdnn4": "gemm_f32_kernel<3>", "tdnn5": "gemm_f32_kernel<3>"})
        return sym.get(tag, "convgemm_kernel<128, false, 3>"), "mfma", PEAK_F32_MATRIX_TFLOPS, "TFLOP/s"
    if pre and tag == "seg_mlp" and xenv("DZ_MLP_HEAD", "1") != "0":
        return "mlp_head_kernel", "mfma", PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS, "TFLOP/s"
    # the first layer behind each SincNet reads planes a one-off norm + split pass wrote (round 6) unless the experiments
    # build's switches take the fused norms away
    nsplit = pre and xenv("DZ_CONV_POOL", "1") != "0" and xenv("DZ_FUSED_NORM", "1") != "0"
    if pre and (tag in ("tdnn2", "tdnn3", "tdnn4", "tdnn5", "lstm_proj", "seg_mlp") or (nsplit and tag in ("lstm_proj0", "tdnn1"))):
        ilv = "true" if xenv("DZ_GP_LOOP", "1") != "0" else "false"
        kern = lambda epi: f"gemm_pre_kernel<{epi}, {ilv}>"
        if gemm_generation() == 2:                                    # k_gemm_g2.hip
            kern = lambda epi: f"gemm_g2_kernel<{epi}, {os.environ.get('DZ_G2_MT', '2')}, 0>"
        if gemm_generation() == 3 and tag in ("tdnn2", "tdnn3"):      # k_gemm_g3.hip: layers with K >= DZ_G3_MINK (1024)
            kern = lambda epi: f"gemm_g3_kernel<{epi}, {os.environ.get('DZ_G3_MT', '4')}>"
        sym = {"lstm_proj": kern(0), "lstm_proj0": kern(0), "seg_mlp": kern(1),
               "tdnn5": "gemm_pre_pool_kernel" if fused_pool else kern(3)}.get(tag, kern(3))
        return sym, "mfma", PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS, "TFLOP/s"
    if tag in ("conv1_pool", "conv2_pool") and xenv("DZ_CONV_POOL", "1") != "0":
        return ("conv_pool_h_kernel<80>" if tag == "conv1_pool" else "conv_pool_h_kernel<64>"), "mfma", \
            PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS, "TFLOP/s"
    wm = "2, 1" if xenv("DZ_SPLIT_WM", "4") == "2" else "4, 2"      # norm-on-load layers
    sym = {"conv1_pool": "gemm_split_kernel<3, 1, true, 4>", "conv2_pool": "gemm_split_kernel<3, 1, true, 4>",
           "lstm_proj": "gemm_split_kernel<4, 2, false, 0>", "lstm_proj0": f"gemm_split_kernel<{wm}, true, 0>",
           "seg_mlp": "gemm_split_kernel<4, 2, false, 1>", "tdnn1": f"gemm_split_kernel<{wm}, true, 3>",
           "seg_head": "seg_head_kernel"}
    return sym.get(tag, "gemm_split_kernel<4, 2, false, 3>"), "mfma", PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS, "TFLOP/s"


_T0 = time.monotonic()


def log(msg):
    print(f"[bench +{time.monotonic() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def emit(full, details_path):
    """Full record -> side file + stderr; ONE compact line (<= 4000 characters) -> stdout, nothing after it."""
    from diart_amd import benchline
    where = benchline.write_details(full, details_path) if details_path else None
    print("[bench details] " + json.dumps(full), file=sys.stderr, flush=True)
    sys.stdout.write(benchline.line(full, where) + "\n")
    sys.stdout.flush()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--streams", type=int, default=64, help="concurrent streams per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rehearsal", action="store_true", help="skip the 8-rank host-contention rehearsal (N = 1 only)")
    ap.add_argument("--power-seconds", type=float, default=float(os.environ.get("DZ_BENCH_POWER_S", "1.5")),
                    help="N = 1: length of the untimed pass behind the timed region during which the card's hwmon files "
                         "(package power, shader clock) are sampled -> `power` (0: off)")
    ap.add_argument("--no-tail", action="store_true",
                    help="stop after clustering (skip the C++ aggregation + binarisation tail)")
    ap.add_argument("--cpu-chunks", type=int, default=32, help="windows per batch of the bounded CPU sample (Benchmark's batch_size, inference.py:275)")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=8, help=argparse.SUPPRESS)
    ap.add_argument("--kernel-table", type=str, default="", help="write the per-kernel table here")
    ap.add_argument("--precision", type=str, default="", help="f16x3 | f32 (default: f16x3)")
    ap.add_argument("--recurrence", type=str, default="", help="StreamBatch(recurrence=): valu | 0 | 3 | 4 (default: the engine's choice)")
    ap.add_argument("--lanes", type=int, default=0, help="StreamBatch(lanes=) (default: the engine's choice)")
    ap.add_argument("--inflight", type=int, default=0, help="StreamBatch(inflight=) (default: lanes + 1)")
    ap.add_argument("--no-host-pass", action="store_true",
                    help="skip the extra pass that uploads each step's new audio from pinned host memory")
    ap.add_argument("--pmc", choices=["on", "all", "off"], default=os.environ.get("DZ_BENCH_PMC", "on"),
                    help="N = 1: measure HBM traffic / matrix-core busy per kernel LIVE with rocprofv3 --pmc child "
                         "passes of this command (on: headline precision, ~2 min; all: also the exact-f32 pass; "
                         "off: use the committed passes under profiles/)")
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3, 5],
                    help="BASELINE.json configs index + 1: 2 = 64 streams, pyannote/segmentation + pyannote/embedding "
                         "(the metric's config, default); 3 = segmentation-3.0 (powerset) + ECAPA-TDNN through the "
                         "blocks pipeline, batches of 32 consecutive windows; 5 = VoiceActivityDetection, 250 ms step, batch 1: "
                         "p50 / p95 per-chunk latency; 1 = Benchmark on one 30 s WAV")
    ap.add_argument("--details", type=str, default=os.environ.get("DZ_BENCH_DETAILS", "gpurun_out/bench_details.json"),
                    help="side file for the full record (per-kernel tables, exact-f32 tables, rehearsal, notes); "
                         "stdout carries ONE compact JSON line (diart_amd/benchline.py)")
    ap.add_argument("--serial-steps", type=int, default=8,
                    help="steps of the serialised roofline pass (one lane, one HIP stream, every launch bracketed: a kernel's "
                         "duration there is its alone-time, what rocprofv3 --kernel-trace --stats of the same pass reports); 0 = off")
    ap.add_argument("--overlap-brackets", action="store_true",
                    help="also bracket every 10th step of the TIMED passes (per-kernel durations while the lanes overlap -> "
                         "`roofline_kernels_overlapped` in the details file, --kernel-table); off by default: the roofline comes "
                         "from the serialised pass, and event pairs on a timed step cost it ~15 %")
    ap.add_argument("--serial-only", action="store_true",
                    help="run ONLY the serialised roofline pass (the command profiles/*_rocprofv3_kernel_stats_serial_*.csv are traces of)")
    ap.add_argument("--no-exact-f32", action="store_true",
                    help="skip the second, untimed-for-`value` pass on the exact-f32 MFMA path")
    return ap.parse_args()


def kernel_table(lib, precision="f16x3"):
    """Per launch tag: launches, total / average duration and the number of chunks those launches
    worked on (recorded by the library next to every bracketed launch: a 32-chunk sub-batch launch
    counts 32), hence algorithmic FLOP and HBM bytes PER LAUNCH at the launch's real batch."""
    rows = []
    name, ms, n, ch = C.c_char_p(), C.c_double(), C.c_longlong(), C.c_longlong()
    for tag in range(32):
        if lib.dz_prof_get(tag, C.byref(name), C.byref(ms), C.byref(n), C.byref(ch)) != 0:
            continue
        if n.value == 0:
            continue
        nm = name.value.decode()
        avg_ms = ms.value / n.value
        cpl = ch.value / n.value
        row = {"kernel": nm, "launches": n.value, "total_ms": round(ms.value, 3), "avg_us": round(avg_ms * 1e3, 2),
               "chunks_per_launch": round(cpl, 2)}
        k = kernels_for(precision).get(nm)
        if k:
            row["alg_gflop_per_launch"] = round(2.0 * k["mac"] * cpl / 1e9, 3)
            row["alg_bytes_per_launch"] = int(k["io"] * cpl + k["w"])
            row["tflops"] = round(2.0 * k["mac"] * cpl / (avg_ms * 1e-3) / 1e12, 2)
            row["gbps"] = round((k["io"] * cpl + k["w"]) / (avg_ms * 1e-3) / 1e9, 1)
            if nm in EXECUTED_MAC:   # the kernel does less arithmetic than the textbook form
                row["executed_gflop_per_launch"] = round(2.0 * EXECUTED_MAC[nm] * cpl / 1e9, 3)
        rows.append(row)
    return rows


def _usable_cores():
    from diart_amd.hostinfo import usable_cores
    return usable_cores()


def cpu_baseline_worker(n_chunks, threads, budget_s):
    """Runs in a CHILD process that never touches the GPU (``bench.py --cpu-worker``): the CPU restatement of the
    reference path (oracle/: "port") on the host cores, shaped like the reference's own ``Benchmark``
    (/root/reference/src/diart/inference.py:275, :392-432): batches of ``n_chunks`` = 32 CONSECUTIVE windows of one
    file through segmentation -> OverlappedSpeechPenalty -> the embedding network on K = 3 repeated waveforms per
    chunk (blocks/embedding.py:57-61) -> normalisation, then per chunk, in order: incremental clustering,
    DelayedAggregation and Binarize (blocks/diarization.py:193-232).  The reference's own classes cannot be loaded
    here (/root/reference does not exist on the GPU box); tools/cpu_reference_baseline.py times THEM around the same
    restated networks in the build container (profiles/r04_cpu_reference_baseline_buildbox.json)."""
    torch.set_num_threads(threads)
    from oracle.models_ref import PyanNetRef, XVectorSincNetRef
    from oracle.functional_ref import overlapped_speech_penalty_ref, normalize_embeddings_ref
    from oracle.clustering_ref import OnlineSpeakerClusteringRef
    from oracle.pyannote_stub import SlidingWindow as SW, SlidingWindowFeature as SWF
    from oracle.tail_ref import TailRef
    from diart_amd.synth import sliding_chunks, synth_embedding_state, synth_segmentation_state, synth_stream
    seg_m, emb_m = PyanNetRef().eval(), XVectorSincNetRef().eval()
    seg_m.load_state_dict(synth_segmentation_state())
    emb_m.load_state_dict(synth_embedding_state())
    B = n_chunks
    stream = synth_stream(0, 5.0 + 0.5 * (4 * B - 1))                 # 4 batches of B consecutive windows
    windows = torch.from_numpy(np.ascontiguousarray(sliding_chunks(stream)))[:, None, :]
    nb = windows.shape[0] // B
    state = {"clu": None, "tail": None, "pos": 0}

    def reset():
        state["clu"], state["tail"], state["pos"] = OnlineSpeakerClusteringRef(0.6, 0.3, 1.0, "cosine", 20), TailRef(0.5, 0.5, 0.6), 0

    def one(dedup=False):
        if state["pos"] >= nb:
            reset()
        i0 = state["pos"] * B
        x = windows[i0:i0 + B]
        with torch.no_grad():
            seg = seg_m(x)
            w = overlapped_speech_penalty_ref(seg)
            if dedup:
                emb = emb_m.forward_multi(x, w)
            else:
                rows = x.repeat(1, 3, 1).reshape(B * 3, 1, -1)
                emb = emb_m(rows, w.permute(0, 2, 1).reshape(B * 3, -1)).view(B, 3, -1)
            emb = normalize_embeddings_ref(emb)
        res = 5.0 / seg.shape[1]
        for j in range(B):
            scores, _ = state["clu"](seg[j].numpy(), emb[j].numpy())
            state["tail"](SWF(scores, SW(start=(i0 + j) * 0.5, duration=res, step=res)))
        state["pos"] += 1

    reset()
    t0 = time.monotonic()
    one()  # warm-up
    warm = time.monotonic() - t0
    t0 = time.monotonic()
    reps = 0
    while reps < 1 or (time.monotonic() - t0 < budget_s and reps < 400):
        one()
        reps += 1
        if warm > budget_s:
            break
    dt = time.monotonic() - t0
    cps = reps * B / dt
    reset()
    t1 = time.monotonic()
    nd = 0
    while nd < 1 or (time.monotonic() - t1 < budget_s / 3 and nd < 100):
        one(dedup=True)
        nd += 1
    cps_dedup = nd * B / (time.monotonic() - t1)
    print(json.dumps({
        "value": round(cps / 2, 3), "unit": "xRT 16 kHz streams (chunks/s / 2)", "cores": threads,
        "kind": "port", "batch": B, "repeats": reps, "dedup_value": round(cps_dedup / 2, 3), "dedup_repeats": nd,
        "sample": f"{reps} Benchmark-shaped batches of {B} consecutive windows (5 s window, 0.5 s step) of one synthetic "
                  f"file: torch-CPU fp32 restatement (oracle/) with the same seeded weights, embedding network on 3 "
                  f"repeated waveforms per chunk as the reference does (blocks/embedding.py:57), then clustering + "
                  f"aggregation + binarisation chunk by chunk; {dt:.1f} s of CPU work after a {warm:.1f} s warm-up "
                  f"batch; de-duplicated embedding (one pass per chunk, {nd} batches): {cps_dedup / 2:.3f} xRT"}), flush=True)


def cpu_baseline(n_chunks):
    """Bounded CPU leg: child process, hard timeout, so the bench line always appears."""
    import subprocess
    usable = _usable_cores()
    threads = max(1, min(usable, int(os.environ.get("DZ_CPU_THREADS", "16"))))
    cmd = [sys.executable, str(ROOT / "bench.py"), "--cpu-worker", "--cpu-chunks", str(n_chunks),
           "--cpu-threads", str(threads)]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads),
               HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    log(f"cpu baseline: {threads} threads of {usable} usable cores ({os.cpu_count()} logical)")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=150, env=env)
    except subprocess.TimeoutExpired:
        log("cpu baseline: child exceeded 150 s, killed")
        return {"value": None, "unit": "xRT 16 kHz streams (chunks/s / 2)", "cores": threads,
                "kind": "port", "sample": "timed out after 150 s"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        log("cpu baseline failed: " + r.stderr[-400:])
        return {"value": None, "unit": "xRT 16 kHz streams (chunks/s / 2)", "cores": threads,
                "kind": "port", "sample": "child failed: " + r.stderr[-200:]}
    return json.loads(lines[-1])


def host_rehearsal(args, precision, usable, ranks=8):
    """What 8 ranks on this node's host cores would leave each rank: the same job (short, headline pass only) as
    a child process pinned to usable / 8 cores, and unpinned for comparison — the 1 -> 8 GPU target is decided by
    16 host cores shared by 8 ranks as much as by the GPUs, and a one-GPU box cannot show it any other way."""
    import subprocess
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    share = max(1, min(len(allowed), usable) // ranks)
    cores = allowed[:share]
    steps = max(args.steps, 100)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", str(steps), "--warmup", "5", "--no-cpu-baseline", "--no-exact-f32",
           "--no-host-pass", "--pmc", "off", "--no-rehearsal", "--power-seconds", "0", "--precision", precision, "--streams", str(args.streams),
           *engine_args(args)]
    res = {}
    for tag, pin in (("unpinned", None), ("pinned", cores)):
        env = dict(os.environ)      # (a pinned child sees < 4 usable cores: StreamBatch stops its pool spinning itself)
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=env,
                               preexec_fn=(lambda c=pin: os.sched_setaffinity(0, c)) if pin is not None else None)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not lines:
                raise RuntimeError("no bench line; stderr: " + r.stderr[-300:])
            line = lines[-1]
            d = json.loads(line)
            res[tag] = {"value": d["value"], "ms_per_step": d["ms_per_step"], "host": d.get("host")}
        except Exception as exc:          # noqa: BLE001 — the bench line must appear whatever happens here
            res[tag] = {"value": None, "error": repr(exc)[:200]}
    ok = res["pinned"].get("value") and res["unpinned"].get("value")
    return {"ranks_emulated": ranks, "cores_per_rank": share, "cores": cores, "steps": steps,
            "pinned": res["pinned"], "unpinned": res["unpinned"],
            "pinned_over_unpinned": round(res["pinned"]["value"] / res["unpinned"]["value"], 4) if ok else None,
            "note": "child runs of this command (headline pass only, no per-kernel brackets): `pinned` may use "
                    "usable_cores / 8 cores (sched_setaffinity), as one of 8 ranks on this node would; target >= 0.95"}


def _match_pmc(files, groups, key):
    """{device kernel symbol -> value} from a per-kernel PMC summary ({"kernels": {full name: {...}}})."""
    got = {}
    for name, v in files.get("kernels", {}).items():
        for g in groups:
            if g.split(" (")[0] in name and v.get(key) is not None:
                got[g] = v[key]
    return got


def engine_args(args):
    """The engine parameters of this command, for its child runs."""
    out = []
    for flag, v in (("--recurrence", args.recurrence), ("--lanes", args.lanes), ("--inflight", args.inflight)):
        if v:
            out += [flag, str(v)]
    return out


def pmc_live(precision, args=None):
    """HBM traffic and matrix-core busy time of every kernel FROM THIS RUN'S BOX: rocprofv3 --pmc
    passes of this same command (short: 3 steps) as child processes, collected as
    MI355X_MICROARCH.md prescribes — separate passes (FETCH_SIZE / WRITE_SIZE / matrix-core busy),
    --kernel-trace only, FETCH_SIZE doubled (gfx950 half-count) — each under a hard timeout.
    Returns {"traffic": {...}, "mfma": {...}, "source": "..."} or None (no rocprofv3, a pass failed,
    or the time budget ran out: the committed passes under profiles/ are used instead)."""
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if Path("/opt/rocm/bin/rocprofv3").exists() else None)
    if rocprof is None:
        return None
    t_start, budget = time.monotonic(), float(os.environ.get("DZ_PMC_BUDGET_S", "200"))
    work = Path(tempfile.mkdtemp(prefix="dz_pmc_"))
    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--no-exact-f32", "--no-host-pass", "--pmc", "off", "--no-rehearsal", "--power-seconds", "0", "--precision", precision,
           *(engine_args(args) if args is not None else [])]
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", TMPDIR="/tmp", DZ_SETTLE_STEPS="0")
    passes = {"FETCH_SIZE": ["FETCH_SIZE"], "WRITE_SIZE": ["WRITE_SIZE"],
              "MFMA": ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]}
    for tag, counters in passes.items():
        left = budget - (time.monotonic() - t_start)
        if left < 30:
            log(f"pmc: time budget spent before the {tag} pass")
            return None
        try:
            r = subprocess.run([rocprof, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d",
                                str(work / tag), "-o", "pmc", "--", *cmd], cwd="/tmp", env=env, capture_output=True,
                               text=True, timeout=min(left, 100))
        except subprocess.TimeoutExpired:
            log(f"pmc: {tag} pass timed out")
            return None
        if r.returncode != 0:
            log(f"pmc: {tag} pass failed: " + r.stderr[-300:])
            return None
        log(f"pmc: {tag} pass done (+{time.monotonic() - t_start:.0f}s)")
    out = {}
    try:
        for script, argv, key in (("pmc_summary.py", [work / "FETCH_SIZE", work / "WRITE_SIZE", work / "traffic.json"], "traffic"),
                                  ("mfma_summary.py", [work / "MFMA", work / "mfma.json"], "mfma")):
            r = subprocess.run([sys.executable, str(ROOT / "tools" / script), *map(str, argv)], capture_output=True,
                               text=True, timeout=60)
            if r.returncode != 0:
                log(f"pmc: {script} failed: " + r.stderr[-300:])
                return None
            out[key] = json.loads(Path(argv[-1]).read_text())
        keep = os.environ.get("DZ_PMC_KEEP")          # keep the two per-kernel tables (the committed fall-back of --pmc off)
        if keep:
            Path(keep).mkdir(parents=True, exist_ok=True)
            suffix = "_f32" if precision == "f32" else ""
            (Path(keep) / f"traffic{suffix}.json").write_text(json.dumps(out["traffic"], indent=1))
            (Path(keep) / f"mfma_util{suffix}.json").write_text(json.dumps(out["mfma"], indent=1))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    out["source"] = (f"live: rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES, separate runs, "
                     f"--kernel-trace only) of `bench.py --steps 3 --precision {precision}` on this box during this run; "
                     "bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 half-count correction)")
    return out


def build_roofline(table, precision, n_sampled, pmc):
    """Per DEVICE kernel: achieved vs the CHIP peak of its binding resource -> (dominant, all)."""
    groups = {}
    for r in table:
        if r["kernel"] not in KERNELS:
            continue
        sym, bound, peak, unit = device_kernel(r["kernel"], precision)
        g = groups.setdefault(sym, {"ms": 0.0, "launches": 0, "gflop": 0.0, "bytes": 0.0, "chunks": 0.0,
                                    "tags": [], "bound": bound, "peak": peak, "unit": unit})
        g["ms"] += r["total_ms"]
        g["launches"] += r["launches"]
        g["gflop"] += r["alg_gflop_per_launch"] * r["launches"]
        g["bytes"] += float(r["alg_bytes_per_launch"]) * r["launches"]
        g["chunks"] += r["chunks_per_launch"] * r["launches"]
        g["tags"].append(r["kernel"])
    if pmc is not None:
        traffic_of = _match_pmc(pmc["traffic"], groups, "hbm_bytes_per_launch")
        mfma_util_of = _match_pmc(pmc["mfma"], groups, "mfma_util")
        busy_of, busy_n = _match_pmc(pmc["mfma"], groups, "mfma_busy_cycles"), _match_pmc(pmc["mfma"], groups, "launches")
        source = pmc["source"]
    else:
        # fall-back: the committed passes of the same command (profiles/, named per round in README.md)
        suffix = "_f32" if precision == "f32" else ""
        tfile, mfile = ROOT / "profiles" / f"traffic{suffix}.json", ROOT / "profiles" / f"mfma_util{suffix}.json"
        traffic_of = _match_pmc(json.loads(tfile.read_text()), groups, "hbm_bytes_per_launch") if tfile.exists() else {}
        mfma_util_of = _match_pmc(json.loads(mfile.read_text()), groups, "mfma_util") if mfile.exists() else {}
        busy_of = _match_pmc(json.loads(mfile.read_text()), groups, "mfma_busy_cycles") if mfile.exists() else {}
        busy_n = _match_pmc(json.loads(mfile.read_text()), groups, "launches") if mfile.exists() else {}
        source = (f"committed: profiles/{tfile.name} / {mfile.name} (rocprofv3 --pmc passes of `bench.py --steps 3` from an "
                  "earlier visit; NOT measured in this run)") if traffic_of else None
    total_ms = sum(v["ms"] for v in groups.values()) or 1.0

    def entry(g, v):
        if v["bound"] == "hbm":
            ach = v["bytes"] / v["ms"] / 1e6          # bytes / ms -> GB/s
        else:
            ach = v["gflop"] / v["ms"]                # GFLOP / ms = TFLOP/s
        e = {"kernel": g, "layers": v["tags"], "bound": v["bound"], "achieved": round(ach, 2),
             "peak": round(v["peak"], 1), "unit": v["unit"], "frac": round(ach / v["peak"], 4),
             "avg_launch_us": round(1e3 * v["ms"] / v["launches"], 2),
             "chunks_per_launch": round(v["chunks"] / v["launches"], 2),
             "launches_per_step": round(v["launches"] / max(1, n_sampled), 2),
             "share_of_kernel_time": round(v["ms"] / total_ms, 4),
             "alg_gflop_per_launch": round(v["gflop"] / v["launches"], 3),
             "alg_bytes_per_launch": int(v["bytes"] / v["launches"]),
             "traffic": traffic_of.get(g), "mfma_util_pmc": mfma_util_of.get(g),
             # SQ_VALU_MFMA_BUSY_CYCLES of one launch, summed over the chip's SIMDs (PMC pass)
             "mfma_busy_cycles_per_launch": (round(busy_of[g] / busy_n[g]) if busy_of.get(g) is not None and busy_n.get(g) else None)}
        if e["traffic"]:
            e["traffic_over_alg_bytes"] = round(e["traffic"] / max(1, e["alg_bytes_per_launch"]), 2)
        if g.startswith("lstm_rec_kernel"):
            cus = min(256.0, 2.0 * v["chunks"] / v["launches"] / lstm_chains_per_wg())   # (chunk[s], direction) per CU
            e["cus_occupied"] = cus
            e["frac_of_occupied_cus_plain_fma"] = round(ach / (PEAK_F32_VECTOR_TFLOPS / 2 * cus / 256.0), 4)
        if g.startswith("lstm_mfma"):
            e["cus_occupied"] = min(256.0, 2.0 * -(-v["chunks"] / v["launches"] // 16))  # 16 chains per workgroup
            e["frac_of_occupied_cus"] = round(ach / (v["peak"] * e["cus_occupied"] / 256.0), 4)
        return e

    per_kernel = [entry(g, v) for g, v in sorted(groups.items(), key=lambda kv: -kv[1]["ms"])]
    if not per_kernel:            # (no brackets were taken on this pass)
        return None, []
    # share of the chip's CU-time (total time x the fraction of the 256 CUs the launches occupy): reported next to
    # the share of kernel time, never used to choose
    cu_time = {e["kernel"]: groups[e["kernel"]]["ms"] * min(1.0, e.get("cus_occupied", 256.0) / 256.0) for e in per_kernel}
    tot_cu = sum(cu_time.values()) or 1.0
    for e in per_kernel:
        e["share_of_cu_time"] = round(cu_time[e["kernel"]] / tot_cu, 4)
    # The dominant kernel = the device kernel with the largest TOTAL duration, exactly the first row of
    # `rocprofv3 --kernel-trace --stats` of the same pass (the recurrence: four launches per step on eight CUs).
    roof = dict(per_kernel[0])
    roof["peak_note"] = {
        "mfma": ("f16 matrix peak 2500 TFLOP/s / 3: the split-f16 path spends three f16 MFMAs per algorithmic "
                 "product; `achieved` counts algorithmic FLOPs only" if roof["peak"] > 200 else "exact-f32 matrix peak"),
        "valu": "chip f32 vector peak (256 CUs); the kernel occupies `cus_occupied` CUs, one latency-bound chain each",
        "hbm": "HBM3E peak"}[roof["bound"]]
    roof["traffic_source"] = source
    return roof, per_kernel


def step_level(per_kernel, ms_per_step, precision, source):
    """What the north-star asks for the STEP: matrix-core utilisation and HBM GB/s against the gfx950 peaks —
    (mfma_util_step, hbm_gbps_step, the dominant MATRIX kernel's roofline entry).  Matrix-core busy time and HBM
    bytes per launch come from the rocprofv3 --pmc passes (`source`), launches per step and durations from this
    run's own brackets."""
    step_us = 1e3 * ms_per_step
    # matrix-pipe busy SIMD-cycles of a step (PMC, per launch x launches per step) over the SIMD-cycles the step
    # offers at the nominal 2.4 GHz (256 CUs x 4 SIMDs): a LOWER bound of the utilisation when the chip clocks down
    busy = sum((k.get("mfma_busy_cycles_per_launch") or 0) * k["launches_per_step"] for k in per_kernel)
    have_util = any(k.get("mfma_busy_cycles_per_launch") for k in per_kernel)
    busy_us = busy / (1024 * 2400.0)                 # -> microseconds of ALL SIMDs busy
    gflop = sum(k["alg_gflop_per_launch"] * k["launches_per_step"] for k in per_kernel if k["bound"] == "mfma")
    products = 1 if precision == "f32" else SPLIT_PRODUCTS
    peak = PEAK_F32_MATRIX_TFLOPS if precision == "f32" else PEAK_F16_MATRIX_TFLOPS
    mfma = {"busy_frac_pmc": round(busy_us / step_us, 4) if have_util else None,
            "issued_tflops": round(products * gflop / ms_per_step, 1),       # GFLOP / ms = TFLOP/s
            "peak_tflops": peak, "frac_of_peak": round(products * gflop / ms_per_step / peak, 4),
            "busy_simd_cycles_per_step": int(busy) if have_util else None,
            "note": ("busy_frac_pmc = SQ_VALU_MFMA_BUSY_CYCLES of the step's launches (PMC pass, per launch x launches per "
                     "step) / (1024 SIMDs x 2.4 GHz x step time); issued_tflops = matrix FLOPs the step issues (%d MFMA product%s per "
                     "algorithmic product) / step time, against the dense %s matrix peak" %
                     (products, "" if products == 1 else "s", "f32" if precision == "f32" else "f16")),
            "source": source}
    byts = sum((k["traffic"] or 0) * k["launches_per_step"] for k in per_kernel)
    alg = sum(k["alg_bytes_per_launch"] * k["launches_per_step"] for k in per_kernel)
    # a kernel whose PMC row was not found would silently drop its bytes: name it, and count its algorithmic bytes
    missing = [k["kernel"] for k in per_kernel if byts and not k["traffic"]]
    byts += sum(k["alg_bytes_per_launch"] * k["launches_per_step"] for k in per_kernel if byts and not k["traffic"])
    hbm = {"gbps": round(byts / step_us / 1e3, 1) if byts else None, "bytes_per_step": int(byts) if byts else None,
           "kernels_without_pmc_row_counted_at_algorithmic_bytes": missing,
           "peak_gbps": PEAK_HBM_GBPS, "frac_of_peak": round(byts / step_us / 1e3 / PEAK_HBM_GBPS, 4) if byts else None,
           "alg_bytes_per_step": int(alg), "source": source}
    mk = next((dict(k) for k in per_kernel if k["bound"] == "mfma" and not k["kernel"].startswith("lstm_")), None)
    return mfma, hbm, mk


# ---- BASELINE.json configs[2]: segmentation-3.0 (powerset) + speechbrain ECAPA-TDNN -------------------
# algorithmic MACs per 5 s chunk: the segmentation trunk as configs[1] (the classifier has 7 outputs) + 3
# ECAPA rows per chunk (reference-shaped call, blocks/embedding.py:56-61: one row per local speaker; the
# mask-selected samples differ per speaker, nothing to de-duplicate): fbank STFT-as-GEMM 498 x 400 x 402,
# block 0 80 -> 1024 k5, three SE-Res2Net blocks (two 1024 x 1024 1 x 1 + seven 128 x 128 k3 + SE), MFA
# 3072 x 3072, attentive pooling 9216 -> 128 -> 3072, fc 6144 -> 192; ~498 frames for a full 5 s mask
ECAPA_MAC_PER_ROW = (498 * (400 * 402 + 201 * 80) + 498 * (80 * 5 * 1024) + 3 * 498 * (2 * 1024 * 1024 + 7 * 128 * 128 * 3 + 2 * 1024 * 128)
                     + 498 * 3072 * 3072 + 498 * (9216 * 128 + 128 * 3072) + 6144 * 192)


def config3(args):
    """`bench.py --config 3`: one step = SpeakerDiarization.__call__ on 32 CONSECUTIVE windows of one
    synthetic stream (Benchmark's batch: inference.py:259-266) with the config-3 models: powerset
    segmentation -> hard multilabel, min-max normalised OSP weights as masks, ECAPA-TDNN on 96 rows,
    clustering + aggregation + binarisation (C++).  Same JSON contract; value = chunks/s / 2."""
    from diart_amd import models as M
    from diart_amd.blocks import SpeakerDiarization, SpeakerDiarizationConfig
    from diart_amd.features import SlidingWindow, SlidingWindowFeature
    from diart_amd.hostinfo import limit_host_threads
    from diart_amd.models import default_precision
    from diart_amd.synth import synth_ecapa_state, synth_segmentation_state, synth_stream
    limit_host_threads()
    if args.gpus != 1:
        raise SystemExit("bench.py --config 3 is a single-GPU line (one pipeline = one stream)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (the HIP path has no CPU fallback)")
    device = torch.device("cuda", 0)
    precision = args.precision or default_precision()
    B = 32
    cfg = SpeakerDiarizationConfig(
        segmentation=M.SegmentationModel.from_state(synth_segmentation_state(seed=77, powerset=True), max_batch=B,
                                                    powerset=True, precision=precision),
        embedding=M.EmbeddingModel.from_state(synth_ecapa_state(), max_batch=3 * B, precision=precision),
        latency=0.5, tau_active=0.5, normalize_embedding_weights=True, device=device)
    pipe = SpeakerDiarization(cfg)
    total = args.steps + args.warmup
    S, H = 80000, 8000
    stream = synth_stream(4242, 5.0 + 0.5 * (B * total + 1))
    chunks = [SlidingWindowFeature(stream[i * H:i * H + S, None], SlidingWindow(start=i * 0.5, duration=1 / 16000, step=1 / 16000))
              for i in range(B * total)]
    log(f"config 3: {len(chunks)} windows of one {len(stream) / 16000:.0f} s stream, batches of {B}, precision {precision}")
    from diart_amd import _lib
    lib = _lib.load()
    for i in range(args.warmup):
        pipe(chunks[i * B:(i + 1) * B])
    torch.cuda.synchronize()
    # the ECAPA rows of a step are padded to the frames of the LONGEST kept row of its batch (the reference pads the
    # same way: PretrainedSpeakerEmbedding's pad_sequence), so the algorithmic work of a step follows the batch
    # geometry of that step, not a full 5 s mask: read back (one ctypes call, no synchronisation) after every step
    ecapa = cfg.embedding.model
    frames_of = lambda: ecapa.last_frames(S)           # noqa: E731
    t0 = time.perf_counter()
    turns, frames_timed = 0, []
    for i in range(args.warmup, total):
        turns += sum(len(a) for a, _ in pipe(chunks[i * B:(i + 1) * B]))
        frames_timed.append(frames_of())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    cps = B * args.steps / elapsed
    T_FULL = 498                                         # frames of a full 5 s mask (ECAPA_MAC_PER_ROW is written for it)
    t_timed = sum(frames_timed) / max(1, len(frames_timed))
    gflop_chunk = 2.0 * (656_230_928 + 3 * ECAPA_MAC_PER_ROW * t_timed / T_FULL) / 1e9
    # ---- where a step's wall time goes (a few extra untimed steps; every phase of the synchronous blocks API ends
    # in a `.cpu()`, so host clocks around the three calls of SpeakerDiarization.__call__ are honest) -------------
    phases = {"stack": 0.0, "segmentation": 0.0, "embedding": 0.0, "finalise": 0.0}
    nph = min(4, args.steps)
    for i in range(nph):
        w = chunks[(args.warmup + i) * B:(args.warmup + i + 1) * B]
        t0 = time.perf_counter()
        batch = torch.stack([torch.from_numpy(c.data) for c in w])
        t1 = time.perf_counter()
        seg_ = pipe.segmentation(batch)
        t2 = time.perf_counter()
        emb_ = pipe.embedding(batch, seg_)
        t3 = time.perf_counter()
        pipe.finalise(w, seg_, emb_)
        t4 = time.perf_counter()
        for k, v in zip(phases, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            phases[k] += 1e3 * v / nph
    phases = {k: round(v, 3) for k, v in phases.items()}
    log(f"config 3 phases, ms per step: {phases}")
    # ---- per-kernel brackets (the dispatches' own timestamps), on a few extra steps after the timed region ------
    nprof = min(4, args.steps)
    lib.dz_prof_enable(1)
    frames_prof = []
    for i in range(nprof):
        pipe(chunks[(args.warmup + i) * B:(args.warmup + i + 1) * B])
        frames_prof.append(frames_of())
    lib.dz_prof_collect()
    name, ms, n_, ch = C.c_char_p(), C.c_double(), C.c_longlong(), C.c_longlong()
    split = precision != "f32"
    T_ROW = sum(frames_prof) / max(1, len(frames_prof))  # frames every row of the bracketed steps was padded to (mean)
    mac_row = {                                          # algorithmic MACs per ECAPA row, by bracket
        "ecapa_fbank": T_ROW * (400 * 402 + 201 * 80), "ecapa_block0": T_ROW * 80 * 5 * 1024,
        "ecapa_wide1x1": T_ROW * (6 * 1024 * 1024 + 3072 * 3072), "ecapa_res2net": 3 * T_ROW * 7 * 128 * 128 * 3,
        "ecapa_se": 3 * 2 * 1024 * 128, "ecapa_asp": 6144 * 128 + T_ROW * (3072 * 128 + 128 * 3072), "ecapa_fc": 6144 * 192}
    groups = []
    for tag in range(32):
        if lib.dz_prof_get(tag, C.byref(name), C.byref(ms), C.byref(n_), C.byref(ch)) != 0 or n_.value == 0:
            continue
        nm = name.value.decode()
        per_step_ms = ms.value / nprof
        g = {"kernel": nm, "launches_per_step": round(n_.value / nprof, 2), "ms_per_step": round(per_step_ms, 3),
             "avg_launch_us": round(1e3 * ms.value / n_.value, 1)}
        if nm in mac_row:
            gf = 2.0 * mac_row[nm] * 3 * B / 1e9                # 3 rows per chunk, B chunks per step
            on_f16 = split and nm in ("ecapa_wide1x1", "ecapa_block0", "ecapa_fbank", "ecapa_res2net", "ecapa_asp")
            peak = PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS if on_f16 else PEAK_F32_MATRIX_TFLOPS
            g.update({"bound": "mfma", "alg_gflop_per_step": round(gf, 1), "achieved": round(gf / per_step_ms, 2),
                      "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(gf / per_step_ms / peak, 4)})
        else:
            k = kernels_for(precision).get(nm)
            if k and k["mac"]:
                gf = 2.0 * k["mac"] * B / 1e9
                g.update({"alg_gflop_per_step": round(gf, 2), "achieved": round(gf / per_step_ms, 2), "unit": "TFLOP/s"})
        groups.append(g)
    lib.dz_prof_enable(0)
    tot = sum(g["ms_per_step"] for g in groups) or 1.0
    for g in groups:
        g["share_of_kernel_time"] = round(g["ms_per_step"] / tot, 4)
    groups.sort(key=lambda g: -g["ms_per_step"])
    dom = next((g for g in groups if g.get("bound") == "mfma"), groups[0] if groups else {})
    out = {
        "metric": "real-time-factor xRT streams/GPU @500ms step", "value": round(cps / 2, 2),
        "unit": "xRT 16 kHz streams (chunks/s / 2)", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if precision == "f32" else "f16x3", "data": "synthetic",
        "config": {"workload": "configs[2]: single MI355X, pyannote/segmentation-3.0 (powerset) + speechbrain ECAPA-TDNN "
                               "architectures (random-init weights), 5 s window / 500 ms step, one synthetic stream through "
                               "the blocks pipeline in batches of 32 consecutive windows (96 embedding rows per step)",
                   "chunks_per_step": B, "speech_turns_emitted": turns, "host_phases_ms": phases,
                   "ecapa_frames_per_row": {"timed_mean": round(t_timed, 1), "timed_min": min(frames_timed, default=0),
                                            "timed_max": max(frames_timed, default=0), "bracketed_mean": round(T_ROW, 1),
                                            "full_mask": T_FULL}},
        "roofline": dict(dom, traffic=None, whole_path_tflops=round(cps * gflop_chunk / 1e3, 2),
                         alg_gflop_per_chunk=round(gflop_chunk, 2), kernel_time_ms_per_step=round(tot, 3),
                         peak_note="f16 matrix peak / 3 for the layers on the split-f16 kernels (three MFMAs per "
                                   "algorithmic product), exact-f32 matrix peak for the others; brackets = the "
                                   "dispatches' own timestamps over %d steps after the timed region; algorithmic work "
                                   "= 96 rows x the frames the step's batch was padded to (config.ecapa_frames_per_row), "
                                   "not a full 5 s mask" % nprof),
        "roofline_kernels": groups,
        "cpu_baseline": None,
    }
    emit(out, args.details)


def config5(args):
    """`bench.py --config 5`: BASELINE.json configs[4] — the VoiceActivityDetection pipeline (segmentation-only path,
    /root/reference/src/diart/blocks/vad.py:127-191), 250 ms step, ONE stream, batch 1: per-chunk latency of
    `pipeline([chunk])` clocked like the reference's Chronometer (utils.py:13-43: a host clock around the call, the
    chunk handed over as host memory, H2D and D2H inside).  --steps = timed chunks (at least 200), --warmup = untimed
    ones.  `value` = p50 in ms (lower is better); reference points of the README (:169): 12 ms CPU, 8 ms RTX 4060."""
    from diart_amd import models as M
    from diart_amd.blocks import VoiceActivityDetection, VoiceActivityDetectionConfig
    from diart_amd.features import SlidingWindow, SlidingWindowFeature
    from diart_amd.hostinfo import limit_host_threads
    from diart_amd.models import default_precision
    from diart_amd.synth import synth_segmentation_state, synth_stream
    limit_host_threads()
    if args.gpus != 1:
        raise SystemExit("bench.py --config 5 is a single-GPU, single-stream line")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (the HIP path has no CPU fallback)")
    device = torch.device("cuda", 0)
    precision = args.precision or default_precision()
    n, warm, step = max(args.steps, 200), max(args.warmup, 5), 0.25
    SR = 16000
    S, H = 5 * SR, int(round(step * SR))
    stream = synth_stream(5, 5.0 + step * (n + warm + 2))
    seg = M.SegmentationModel.from_state(synth_segmentation_state(), max_batch=1, precision=precision)
    pipe = VoiceActivityDetection(VoiceActivityDetectionConfig(segmentation=seg, step=step, device=device))
    times, turns = [], 0
    import gc
    for i in range(n + warm):
        c = SlidingWindowFeature(stream[i * H:i * H + S, None], SlidingWindow(start=i * step, duration=1 / SR, step=1 / SR))
        if i == warm:
            gc.collect()
            gc.freeze()
            torch.cuda.synchronize()
            t_all = time.perf_counter()
        t0 = time.monotonic()
        out = pipe([c])
        times.append(1e3 * (time.monotonic() - t0))
        turns += len(out[0][0])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_all
    t = np.array(times[warm:])
    lat = {"p50": round(float(np.percentile(t, 50)), 3), "p95": round(float(np.percentile(t, 95)), 3),
           "mean": round(float(t.mean()), 3), "max": round(float(t.max()), 3)}
    gflop = 1.312                                    # segmentation only
    out = {
        "metric": "per-chunk latency of VoiceActivityDetection([chunk]), p50", "value": lat["p50"], "unit": "ms",
        "n_gpus": 1, "steps": n, "warmup": warm, "ms_per_step": round(1e3 * elapsed / n, 3), "higher_is_better": False,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32" if precision == "f32" else "f16x3", "data": "synthetic",
        "config": {"workload": "configs[4]: VoiceActivityDetection pipeline (pyannote/segmentation architecture, random-init "
                               "weights), 5 s window / 250 ms step, one synthetic 16 kHz stream, batch 1, chunk handed over "
                               "as host memory (H2D + D2H inside the clocked call)",
                   "chunks_per_step": 1, "speech_turns_emitted": turns,
                   "reference_points_ms": {"cpu": 12, "rtx4060": 8, "source": "reference README.md:169"}},
        "latency_ms": lat, "chunk_ms": 1e3 * step,
        "roofline": {"kernel": "whole segmentation network at batch 1", "bound": "latency",
                     "achieved": round(gflop / lat["p50"], 3), "peak": PEAK_F32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(gflop / lat["p50"] / PEAK_F32_VECTOR_TFLOPS, 5), "traffic": None,
                     "alg_gflop_per_launch": gflop,
                     "note": "one chunk = 4 x 293 dependent recurrence steps on 2 of 256 CUs; latency-bound by construction"},
        "cpu_baseline": None,
    }
    emit(out, args.details)


def config1(args):
    """`bench.py --config 1`: BASELINE.json configs[0] — `diart.benchmark` (the reference's Benchmark class,
    /root/reference/src/diart/inference.py:392-432) on ONE 30 s 16 kHz WAV with SpeakerDiarization,
    pyannote/segmentation + pyannote/embedding architectures, batch_size 32: 51 chunks (inference.py:81-83).  A step =
    one complete run over the file (read WAV -> chunks -> pipeline -> RTTM written); `value` = chunks/s / 2."""
    import tempfile
    from diart_amd import models as M
    from diart_amd.blocks import SpeakerDiarization, SpeakerDiarizationConfig
    from diart_amd.hostinfo import limit_host_threads
    from diart_amd.inference import Benchmark, write_wav
    from diart_amd.models import default_precision
    from diart_amd.synth import synth_embedding_state, synth_segmentation_state, synth_stream
    limit_host_threads()
    if args.gpus != 1:
        raise SystemExit("bench.py --config 1 is a single-GPU line (one file); tools/benchmark_files.py --gpus N shards files")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (the HIP path has no CPU fallback)")
    device = torch.device("cuda", 0)
    precision = args.precision or default_precision()
    seconds, B = 30.0, 32
    work = Path(tempfile.mkdtemp(prefix="dz_config1_"))
    (work / "wav").mkdir()
    write_wav(work / "wav" / "synthetic_00.wav", synth_stream(1000, seconds), 16000)
    cfg = SpeakerDiarizationConfig(
        segmentation=M.SegmentationModel.from_state(synth_segmentation_state(), max_batch=B, precision=precision),
        embedding=M.EmbeddingModel.from_state(synth_embedding_state(), max_batch=B, precision=precision),
        latency=0.5, device=device)
    bench = Benchmark(work / "wav", None, work / "rttm", show_report=False, batch_size=B)
    chunks = int(np.ceil((seconds - 5.0 + 0.5) / 0.5))
    for _ in range(max(1, args.warmup)):
        bench(SpeakerDiarization, cfg)
    torch.cuda.synchronize()
    walls = []
    t_all = time.perf_counter()
    for _ in range(args.steps):
        t0 = time.perf_counter()
        bench(SpeakerDiarization, cfg)
        walls.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_all
    rttm = (work / "rttm" / "synthetic_00.rttm").read_text().splitlines()
    import shutil
    shutil.rmtree(work, ignore_errors=True)
    cps = chunks * args.steps / elapsed
    out = {
        "metric": "real-time-factor xRT streams/GPU @500ms step", "value": round(cps / 2, 2),
        "unit": "xRT 16 kHz streams (chunks/s / 2)", "n_gpus": 1, "steps": args.steps, "warmup": max(1, args.warmup),
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if precision == "f32" else "f16x3", "data": "synthetic",
        "config": {"workload": "configs[0]: Benchmark(SpeakerDiarization) on one 30 s 16 kHz WAV, pyannote/segmentation + "
                               "pyannote/embedding architectures (random-init weights), batch_size 32, latency = step = 0.5 s; "
                               "a step = the whole file (WAV read, 51 chunks, RTTM written)",
                   "chunks_per_step": chunks, "path": bench.last_path, "rttm_lines": len(rttm)},
        "file_seconds": seconds, "wall_s": {"p50": round(float(np.median(walls)), 4), "max": round(float(max(walls)), 4)},
        "roofline": {"kernel": "whole file", "bound": "latency", "achieved": round(cps * ALG_GFLOP_PER_CHUNK / 1e3, 2),
                     "peak": PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS, "unit": "TFLOP/s",
                     "frac": round(cps * ALG_GFLOP_PER_CHUNK / 1e3 / (PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS), 5),
                     "traffic": None, "note": "51 chunks in two batches (32 + 19): launch- and host-bound, not a kernel roofline"},
        "cpu_baseline": None,
    }
    emit(out, args.details)


def main():
    args = parse()
    if os.environ.get("DZ_BENCH_SCHED"):               # experiment: fifo:<prio> | nice:<n> for every thread created from here on
        kind, _, val = os.environ["DZ_BENCH_SCHED"].partition(":")
        try:
            if kind == "fifo":
                os.sched_setscheduler(0, os.SCHED_FIFO, os.sched_param(int(val or 1)))
            elif kind == "rr":
                os.sched_setscheduler(0, os.SCHED_RR, os.sched_param(int(val or 1)))
            elif kind == "nice":
                os.nice(int(val or -10))
            print(f"[bench] scheduling: {kind} {val} applied", file=sys.stderr, flush=True)
        except Exception as exc:      # noqa: BLE001
            print(f"[bench] scheduling: {kind} {val} refused: {exc!r}", file=sys.stderr, flush=True)
    if args.cpu_worker:
        return cpu_baseline_worker(args.cpu_chunks, args.cpu_threads, 20.0)
    if args.config == 3:
        return config3(args)
    if args.config == 5:
        return config5(args)
    if args.config == 1:
        return config1(args)
    from diart_amd import distributed as D
    # `python bench.py --gpus N` as ONE process: start the N ranks ourselves (torch.distributed.run,
    # one rank per GPU, RCCL); under the driver's own torchrun WORLD_SIZE is set and this is a no-op
    rc = D.self_launch(args.gpus, str(ROOT / "bench.py"), sys.argv[1:])
    if rc is not None:
        raise SystemExit(rc)
    from diart_amd import _lib
    from diart_amd.models import HipEmbedding, HipSegmentation, default_precision
    from diart_amd.pipeline import StreamBatch
    from diart_amd.synth import synth_embedding_state, synth_segmentation_state, synth_streams

    precision = args.precision or default_precision()
    log(f"start (precision {precision})")
    from diart_amd.hostinfo import limit_host_threads
    limit_host_threads()
    rank, world, local = D.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU (the HIP path has no CPU fallback)")
    if world > 1:
        log(f"rank {rank}/{world}: process group up, backend {torch.distributed.get_backend()} "
            f"({'RCCL' if torch.distributed.get_backend() == 'nccl' else 'rehearsal'}), "
            f"device {os.environ.get('DZ_FORCE_DEVICE', local)} of {torch.cuda.device_count()}")
        if "DZ_FORCE_DEVICE" not in os.environ and torch.cuda.device_count() < world:
            raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s) visible")
    # DZ_FORCE_DEVICE: every rank on one GPU (single-GPU rehearsal of the multi-rank path, with
    # DZ_DIST_BACKEND=gloo); the driver's real runs use one GPU per rank over RCCL
    device = torch.device("cuda", int(os.environ.get("DZ_FORCE_DEVICE", local)))
    torch.cuda.set_device(device)
    from diart_amd.hostinfo import bind_rank
    affinity = bind_rank(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)),
                         int(os.environ["DZ_FORCE_DEVICE"]) if "DZ_FORCE_DEVICE" in os.environ else None)
    if world > 1:
        log(f"rank {rank}: cpu affinity {affinity}")
    lib = _lib.load()
    # the library's two run-time options, from the environment of THIS command (tests, A/B runs)
    _lib.set_option("pool_fuse", int(os.environ.get("DZ_POOL_FUSE", "1") != "0"))
    _lib.set_option("f32_gemm", int(os.environ.get("DZ_F32_GEMM", "1") != "0"))

    # ---- weights: synthesised on rank 0, broadcast over RCCL ---------------------------
    seg_state = synth_segmentation_state() if rank == 0 or world == 1 else None
    emb_state = synth_embedding_state() if rank == 0 or world == 1 else None
    if world > 1:
        # only rank 0 holds weights; the others know the architecture (key, shape, dtype) and
        # receive the values as one flat buffer over RCCL
        from diart_amd.synth import embedding_spec, segmentation_spec
        seg_state = D.broadcast_state(seg_state, segmentation_spec(), device)
        emb_state = D.broadcast_state(emb_state, embedding_spec(), device)

    # every rank must hold the same weights after the broadcast: a checksum per rank goes into the line
    wsum = float(sum(v.double().abs().sum().item() for v in list(seg_state.values()) + list(emb_state.values())))
    wsums = [w[0] for w in D.gather_counts([wsum], device)] if world > 1 else [wsum]

    # ---- synthetic streams of this rank, resident in HBM -------------------------------
    n = args.streams
    STREAMS_PER_GPU[0] = n
    hop, S = 8000, 80000
    total_steps = args.steps + args.warmup
    seconds = (S + hop * (total_steps + 1)) / 16000.0
    log(f"synthesising {n} streams of {seconds:.1f}s")
    audio_cpu = torch.from_numpy(synth_streams(n, seconds, seed0=rank * n))
    audio = audio_cpu.to(device)
    log("streams resident in HBM")
    assert audio.stride(0) % 4 == 0

    # host threads of the clustering / output tail: sized from the MEASURED host work of a step
    # (below, after the settling steps), not from cores // ranks: at 8 ranks on a 16-core grant that
    # rule left 2 threads for ~1.5 ms of CPU work per 1.3 ms step.  The ranks' host phases are short
    # (15-30 % duty) and not synchronised, so the node's cores are time-shared; a rank may use up to
    # twice its even share as long as the node-wide demand (ranks x CPU-ms per step / step) fits.
    from diart_amd.hostinfo import usable_cores
    usable = usable_cores()
    host_threads = max(1, min(8, usable))
    def make_pipe(prec, serial_recurrence=False):
        """The engine of this run, or (serial_recurrence is not False) the MEASUREMENT engine of the roofline pass:
        one lane, one HIP stream, one step in flight, the recurrence kernel named."""
        if serial_recurrence is not False:
            return StreamBatch(HipSegmentation(seg_state, max_batch=n, precision=prec),
                               HipEmbedding(emb_state, max_batch=n, precision=prec),
                               n, device=device, cluster_threads=host_threads, tail=not args.no_tail,
                               recurrence=serial_recurrence, lanes=1, inflight=1, serial=True, warmup=0)
        p_ = StreamBatch(HipSegmentation(seg_state, max_batch=n, precision=prec),
                         HipEmbedding(emb_state, max_batch=n, precision=prec),
                         n, device=device, cluster_threads=host_threads,
                         tail=not args.no_tail, recurrence=args.recurrence or None, lanes=args.lanes or None,
                         inflight=args.inflight or None)
        if p_.recurrence:
            RECURRENCE[prec] = p_.recurrence
        if os.environ.get("DZ_LAUNCH_TRACE"):
            p_.slow_launches = []
        if os.environ.get("DZ_BENCH_D2H") == "memcpy":       # A/B arm: results by hipMemcpyAsync as before round 6
            p_.d2h_by_kernel = False
        return p_

    if args.serial_only:
        from diart_amd.weights import THROUGHPUT_LSTM_VARIANT
        RECURRENCE[precision] = args.recurrence or (str(THROUGHPUT_LSTM_VARIANT) if n >= 64 and precision == "f16x3" else "valu")
        pipe = make_pipe(precision, RECURRENCE[precision] if precision == "f16x3" else None)
    else:
        pipe = make_pipe(precision)

    def window(t):
        return audio[:, t * hop: t * hop + S]

    host = {"launch": 0.0, "finish": 0.0}

    # every 10th step of the timed region carries the per-kernel event pairs (DZ_PROF_EVERY=1: all)
    PROF_EVERY = max(1, int(os.environ.get("DZ_PROF_EVERY", "10")))
    # brackets on the TIMED passes: only on request (or when the serialised pass is off and something has to feed the roofline)
    BRACKETS = (args.overlap_brackets or args.serial_steps <= 0 or bool(args.kernel_table)) and not os.environ.get("DZ_NO_PROF")
    sampled = [0]
    stamps = []                    # host clock at every launch (the timed passes report the spread of the step period)

    def run(t_first, count, pipe=None, profiled=False, every=None):
        # pipe.max_inflight steps are launched ahead (pipe.depth of them run concurrently, one per lane; the
        # others wait in their lane's streams) while the host runs the clustering + output tail of the oldest
        pipe = pipe or main_pipe[0]
        inflight = []
        for t in range(t_first, t_first + count):
            if profiled:
                on = (t - t_first) % (every or PROF_EVERY) == 0
                lib.dz_prof_pause(0 if on else 1)
                sampled[0] += int(on)
            h0 = time.perf_counter()
            stamps.append(h0)
            inflight.append(pipe.launch(window(t)))
            h1 = time.perf_counter()
            if len(inflight) >= pipe.max_inflight:
                pipe.finish(inflight.pop(0), want_scores=True)
            h2 = time.perf_counter()
            host["launch"] += h1 - h0
            host["finish"] += h2 - h1
            parts.append((h1 - h0, h2 - h1, pipe.host_seconds["wait"], pipe.host_seconds["work"]))
        while inflight:
            pipe.finish(inflight.pop(0), want_scores=True)

    main_pipe = [pipe]
    parts = []                     # per launched step: (launch s, finish s, cumulative wait s, cumulative work s)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # untimed: settle clocks, page in the scratch arenas and both output slots, then the W warm-up
    # steps the contract asks for
    run(0, min(total_steps, 10))
    torch.cuda.synchronize()
    # Everything allocated so far (torch, numpy, the pipeline's Python objects) lives for the whole run: take it
    # out of the cyclic collector's generations, or a full collection stops the launching thread for 5 - 8 ms
    # once per few hundred steps (seen as a GPU idle gap of that length in the dispatch timestamps; the steps
    # themselves create no cycles).  What a service loop does after start-up; nothing is skipped.
    import gc
    gc.collect()
    gc.freeze()
    if os.environ.get("DZ_BENCH_GC") == "0":          # experiment: no cyclic collections at all from here on
        gc.disable()
    # ---- size the host threads from what the second batch of settling steps measures -----------
    n_settle = min(total_steps, 10)
    host["launch"] = host["finish"] = 0.0
    pipe.host_seconds["wait"] = pipe.host_seconds["work"] = 0.0
    run(0, n_settle)
    torch.cuda.synchronize()
    hs0 = pipe.host_seconds
    work_wall_ms = 1e3 * hs0["work"] / n_settle                 # clustering + tail, wall, on host_threads
    step_ms0 = 1e3 * (hs0["work"] + hs0["wait"] + host["launch"]) / n_settle
    cpu_ms = work_wall_ms * host_threads                        # upper bound of the CPU time in it
    # enough threads to keep the host half under ~30 % of the step ...
    need = max(1, int(np.ceil(cpu_ms / max(1e-3, 0.3 * step_ms0))))
    # ... capped by twice this rank's even share of the node's usable cores (and by the pool's 8)
    cap = max(2, min(8, int(np.ceil(2.0 * usable / max(1, world)))))
    if os.environ.get("DZ_HOST_THREADS"):
        need = cap = int(os.environ["DZ_HOST_THREADS"])
    host_threads_used = pipe.set_host_threads(min(need, cap))
    log(f"host work per step {work_wall_ms:.3f} ms wall on {host_threads} threads (<= {cpu_ms:.2f} CPU-ms) of a "
        f"{step_ms0:.2f} ms step; {usable} usable cores, {world} rank(s): {host_threads_used} host threads per rank "
        f"(node-wide demand ~{world * cpu_ms / max(1e-3, step_ms0):.1f} cores)")
    host_threads = host_threads_used
    # The timed region brackets every PROF_EVERY-th step's launches with timing events.  The FIRST dispatches of a
    # stream that carry such events are a start-up cost of the runtime (seen once as a ~11 ms launch call inside a
    # 25 ms timed region: 17 718 xRT on a box whose untouched passes of the same run gave 24 400 - 27 200): pay it here
    def prof_warm(p):
        if BRACKETS:
            lib.dz_prof_enable(1)
            run(0, min(total_steps, 2 * p.max_inflight), p, profiled=True, every=1)      # every lane's streams
            lib.dz_prof_collect()
            lib.dz_prof_enable(0)

    prof_warm(pipe)
    # untimed settling beyond the contract's W warm-up steps: the timed region of the driver's form is ~25 ms
    # (20 steps), so ONE runtime / OS stall of a few ms inside it moves `value` by 20 - 30 %; a process that is
    # 1.2 s old still has such stalls ahead of it (first wrap of the queues' kernarg / signal rings, allocator
    # growth).  DZ_SETTLE_STEPS more steps (default 150, ~0.2 s) — in the same pattern the timed region uses,
    # every PROF_EVERY-th step carrying its event pairs — get them out of the way.  Nothing is skipped or cached:
    # the timed region below is exactly K full steps on windows of its own.
    settle = int(os.environ.get("DZ_SETTLE_STEPS", "150"))
    if settle > 0:
        prof_on = BRACKETS
        lib.dz_prof_enable(1 if prof_on else 0)
        done_ = 0
        while done_ < settle:
            k = min(total_steps, settle - done_)
            run(0, k, pipe, profiled=prof_on)
            done_ += k
            if prof_on:
                lib.dz_prof_collect()
        lib.dz_prof_enable(0)
        torch.cuda.synchronize()
    run(0, args.warmup)
    torch.cuda.synchronize()
    log("warm-up done")
    period = {}                    # per timed pass: launch-to-launch period p50 / max, host launch time per step

    def timed_pass(p, label):
        """K timed steps of pipeline `p` (barrier + synchronize on both sides, max over ranks) with the
        per-kernel event pairs on every PROF_EVERY-th step -> (elapsed s, kernel table, sampled steps)."""
        prof = BRACKETS
        lib.dz_prof_enable(1 if prof else 0)
        sampled[0] = 0
        host["launch"] = host["finish"] = 0.0
        p.host_seconds["wait"] = p.host_seconds["work"] = 0.0
        cpu0 = time.process_time()                     # CPU time of every thread of this process
        del stamps[:]
        del parts[:]
        el = D.timed_max_over_ranks(lambda: run(args.warmup, args.steps, p, profiled=prof), device)
        gaps = np.diff(np.asarray(stamps)) * 1e3 if len(stamps) > 2 else np.zeros(1)
        # launch-to-launch period of the host loop inside the timed region: a one-off stall (runtime, OS) shows up as
        # max >> p50 — `value` is still total / K, as the contract says
        host["step_period_ms"] = {"p50": round(float(np.median(gaps)), 3), "max": round(float(gaps.max()), 3)}
        host["step_gaps_ms"] = [round(float(g), 3) for g in gaps]        # details file only (where a stall sits)
        # ... and what the host was doing in the slow periods: (step, period, launch, waiting for the GPU, clustering / tail) ms
        pw = np.diff(np.asarray([0.0] + [q[2] for q in parts])) if parts else np.zeros(0)
        pk = np.diff(np.asarray([0.0] + [q[3] for q in parts])) if parts else np.zeros(0)
        host["slow_periods_ms"] = [[int(i), round(float(gaps[i]), 2), round(1e3 * parts[i][0], 2), round(1e3 * float(pw[i]), 2),
                                    round(1e3 * float(pk[i]), 2)] for i in range(min(len(gaps), len(parts)))
                                   if gaps[i] > 3.0 * max(0.1, float(np.median(gaps)))][:40]
        host["cpu_ms_per_step"] = 1e3 * (time.process_time() - cpu0) / args.steps
        host["launch_ms_per_step"], host["work_ms_per_step"] = 1e3 * host["launch"] / args.steps, 1e3 * p.host_seconds["work"] / args.steps
        hs = p.host_seconds
        log(f"{label}: {el:.3f}s for {args.steps} steps; host time per step: launch "
            f"{1e3 * host['launch'] / args.steps:.3f} ms, finish {1e3 * host['finish'] / args.steps:.3f} ms = waiting for "
            f"the GPU {1e3 * hs['wait'] / args.steps:.3f} + clustering / tail {1e3 * hs['work'] / args.steps:.3f}")
        lib.dz_prof_collect()
        tab = kernel_table(lib, p.seg.precision)   # read before dz_prof_enable(0) clears the accumulators
        lib.dz_prof_enable(0)
        period[label] = dict(host["step_period_ms"], launch_ms_per_step=round(host["launch_ms_per_step"], 3),
                             gaps_ms=host["step_gaps_ms"], profiled_every=PROF_EVERY if prof else None)
        return el, tab, sampled[0]

    def serial_pass(prec, engine=None):
        """The roofline pass: `--serial-steps` steps on the MEASUREMENT engine (one lane, ONE HIP stream, one step in
        flight: no two kernels overlap), every launch bracketed with its dispatch's own start / stop timestamps ->
        (kernel table, sampled steps, ms per serialised step).  A kernel's duration here is its alone-time at this
        job's shapes: what `rocprofv3 --kernel-trace --stats` of `bench.py --serial-only` reports per kernel
        (profiles/*_rocprofv3_kernel_stats_serial_*.csv; tests/test_roofline_repro.py holds the line to it)."""
        sp = engine or make_pipe(prec, RECURRENCE.get(prec) if prec == "f16x3" else None)
        run(0, min(total_steps, 4), sp)
        torch.cuda.synchronize()
        lib.dz_prof_enable(1)
        sampled[0] = 0
        k = max(1, min(args.serial_steps, total_steps))
        t0 = time.perf_counter()
        run(0, k, sp, profiled=True, every=1)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / k
        lib.dz_prof_collect()
        tab = kernel_table(lib, prec)
        lib.dz_prof_enable(0)
        log(f"serialised roofline pass ({prec}): {k} steps, {ms:.3f} ms per step, {len(tab)} kernel tags")
        return tab, k, ms

    if args.serial_only:
        tab, k, ms = serial_pass(precision, pipe)
        roof, per_kernel = build_roofline(tab, precision, k, None)
        cps = n * 1e3 / ms
        out = {"metric": "real-time-factor xRT streams/GPU @500ms step", "value": round(cps / 2, 2),
               "unit": "xRT 16 kHz streams (chunks/s / 2)", "n_gpus": 1, "steps": k, "warmup": 4, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if precision == "f32" else "f16x3",
               "data": "synthetic",
               "config": {"workload": "configs[1] on the MEASUREMENT engine: one lane, one HIP stream, one step in flight (the "
                                      "serialised roofline pass of bench.py alone; NOT the metric)", "streams_per_gpu": n,
                          "chunks_per_step": n, "lanes": 1, "steps_in_flight": 1, "recurrence": RECURRENCE.get(precision, "valu")},
               "roofline": dict(roof, serialised=True, serialised_ms_per_step=round(ms, 3)),
               "roofline_kernels": per_kernel, "cpu_baseline": None}
        if args.kernel_table:
            Path(args.kernel_table).parent.mkdir(parents=True, exist_ok=True)
            Path(args.kernel_table).write_text(json.dumps(tab, indent=1))
        emit(out, args.details)
        return

    elapsed, table, n_sampled = timed_pass(pipe, f"timed region ({precision})")

    def power_pass(p):
        """What the GPU draws while this pipeline runs: an UNTIMED pass of `--power-seconds` on the same engine, right behind
        the timed region, with the card's own hwmon files (package power, shader clock) sampled every 40 ms by a side thread;
        the median of the last 60 % of the pass (the power reading is itself an average that takes ~0.3 s to settle).
        profiles/r06zl_clock_power.json: the default-precision pipeline sits at the package's power budget."""
        from diart_amd.hwmon import PowerSampler
        sm = PowerSampler(period=0.04, device_index=device.index)
        if not sm.available or (sm.own is None and len(sm.freq) > 1):
            # no hwmon tree, or several cards and none matched by PCI address: another tenant's card could be the one whose
            # power moves most — no reading is better than somebody else's
            sm.stop()
            return None
        try:
            torch.cuda.synchronize()
            time.sleep(1.0)                                  # idle reference (the reading is a ~0.3 s average)
            t_idle = time.time()
            t0 = time.time()
            steps_done, flying = 0, []
            while time.time() - t0 < args.power_seconds:         # one continuous pass: no drain between its steps
                flying.append(p.launch(window(steps_done % total_steps)))
                steps_done += 1
                if len(flying) >= p.max_inflight:
                    p.finish(flying.pop(0), want_scores=True)
            while flying:
                p.finish(flying.pop(0), want_scores=True)
            torch.cuda.synchronize()
            t1 = time.time()
        finally:
            sm.stop()
        card = sm.card()
        _, idle_w, _ = sm.window(t_idle - 0.25, t_idle, card)
        mhz, watts, ns = sm.window(t0 + 0.4 * (t1 - t0), t1 - 0.02, card)
        if watts is None:
            return None
        ms = 1e3 * (t1 - t0) / max(1, steps_done)
        return {"package_w": round(watts), "sclk_mhz": round(mhz), "idle_w": round(idle_w) if idle_w is not None else None,
                "ms_per_step": round(ms, 3), "joules_per_step": round(watts * ms / 1e3, 3), "seconds": round(t1 - t0, 2),
                "steps": steps_done, "samples": ns, "card_matched_by_pci_address": sm.own is not None,
                "source": "amdgpu hwmon (power1_average | power1_input, freq1_input) of the card whose power moved, sampled every "
                          "40 ms by a side thread during an untimed pass behind the timed region; median of its last 60 %"}

    power = None
    if world == 1 and rank == 0 and args.power_seconds > 0:
        try:
            power = power_pass(pipe)
        except Exception as exc:      # noqa: BLE001 — an unreadable hwmon tree must not cost the run its line
            log(f"power pass skipped: {exc!r}")
    if power:
        log(f"power pass: {power['package_w']} W at {power['sclk_mhz']} MHz over {power['steps']} steps "
            f"({power['ms_per_step']} ms / step, {power['joules_per_step']} J / step; idle {power['idle_w']} W)")
    serial = serial_pass(precision) if args.serial_steps > 0 else None
    host_line = {"cpu_ms_per_step": round(host["cpu_ms_per_step"], 3), "launch_ms_per_step": round(host["launch_ms_per_step"], 3),
                 "step_period_ms_in_timed_region": host["step_period_ms"],
                 "clustering_tail_wall_ms_per_step": round(host["work_ms_per_step"], 3), "threads": host_threads,
                 # (details file only) the launch-to-launch gaps of the headline pass: with `steps_in_flight` launched ahead the
                 # host launches in bursts and then waits for the oldest step — max ~ steps_in_flight x the step
                 "step_gaps_ms_in_timed_region": list(host["step_gaps_ms"]),
                 "slow_periods_ms_step_period_launch_wait_work": list(host.get("slow_periods_ms", [])),
                 "slow_launches_ms_by_call": (getattr(pipe, "slow_launches", None) or [])[-20:],
                 "usable_cores": usable,
                 "note": "CPU time of all threads of the rank per step (launching thread + worker pool) in the timed region"}

    # ---- the same job fed from HOST buffers: every step uploads the 500 ms of new audio of each
    # stream (pinned memory -> device ring, dz_ring_push) instead of finding it in HBM ------------
    host_fed = None
    if not args.no_host_pass:
        from diart_amd.pipeline import AudioRing
        ring = AudioRing(n, S, hop, slack_blocks=2 * pipe.max_inflight + 2, device=device)   # >= steps in flight
        blocks = audio_cpu.unfold(1, hop, hop)                     # (n, nblocks, hop) view
        pinned = [blocks[:, i].contiguous().pin_memory() for i in range(S // hop + total_steps)]
        for i in range(S // hop - 1):
            ring.push(pinned[i])

        hmode = int(os.environ.get("DZ_HOSTFED", "7"))        # A/B switches of this pass: 1 = feed stream at high priority,
        #                                                        2 = the block of step t+1 is pushed right behind launch(t), 4 = settling steps
        feed = torch.cuda.Stream(device, priority=-1 if hmode & 1 else 0)     # uploads + the launch's input event: off the default stream

        hf = {"push": 0.0, "launch": 0.0, "finish": 0.0}
        nblk = len(pinned)

        def run_ring(first, count):
            """`count` steps: the new 500 ms of every stream go pinned host -> ring (one scatter kernel that reads the
            pinned block in place), the step is launched on the ring's current window.  With bit 2 the upload of step
            t+1 is enqueued right behind launch(t) — before the host waits for the oldest step — so that it never sits
            at the head of a step's dependent chain (a file / batch feeder has the next block; a live source would push
            on arrival, 500 ms earlier still)."""
            inflight = []
            pushed_ahead = False
            for t in range(first, first + count):
                h0 = time.perf_counter()
                with torch.cuda.stream(feed):
                    if not pushed_ahead:
                        ring.push(pinned[(S // hop - 1 + t) % nblk])
                    h1 = time.perf_counter()
                    inflight.append(pipe.launch(ring))
                    pushed_ahead = False
                    if hmode & 2 and t + 1 < first + count:
                        ring.push(pinned[(S // hop - 1 + t + 1) % nblk])
                        pushed_ahead = True
                h2 = time.perf_counter()
                if len(inflight) >= pipe.max_inflight:
                    pipe.finish(inflight.pop(0), want_scores=True)
                hf["push"] += h1 - h0
                hf["launch"] += h2 - h1
                hf["finish"] += time.perf_counter() - h2
            while inflight:
                pipe.finish(inflight.pop(0), want_scores=True)

        if hmode & 4 and settle > 0:        # the same settling the headline pass gets (its own pattern, untimed)
            done_ = 0
            while done_ < settle:
                k = min(total_steps, settle - done_)
                run_ring(0, k)
                done_ += k
            torch.cuda.synchronize()
            ring.reset()
            pipe.reset()
            for i in range(S // hop - 1):
                ring.push(pinned[i])
        run_ring(0, args.warmup)
        eh = D.timed_max_over_ranks(lambda: run_ring(args.warmup, args.steps), device)
        host_fed = {"value": round(D.whole_job_rate(n, args.steps, eh, world) / 2, 2), "ms_per_step": round(1e3 * eh / args.steps, 3),
                    "h2d_bytes_per_step": n * hop * 4, "mode": hmode, "settle_steps": settle if hmode & 4 else 0,
                    "note": "PCIe-inclusive: per step the 8000 new samples of every stream go pinned host -> "
                            "device ring (dz_ring_push), the window is read in place; not `value`"}
        log(f"host-fed pass: {eh:.3f}s; host time per step: push {1e3 * hf['push'] / (args.steps + args.warmup):.3f} ms, "
            f"launch {1e3 * hf['launch'] / (args.steps + args.warmup):.3f} ms, finish "
            f"{1e3 * hf['finish'] / (args.steps + args.warmup):.3f} ms")

    # ---- the same job on the exact-f32 MFMA path: the number at the reference's own arithmetic,
    # measured the same way (own per-kernel brackets, own roofline), not `value` ----------------------
    exact = None
    if precision != "f32" and not args.no_exact_f32:
        p32 = make_pipe("f32")
        prof_warm(p32)
        # the same settling as the headline pass, a third as many steps (they take twice as long): a timed region that
        # starts right behind the creation of an engine (arenas allocated, weights packed) reads 7 - 8 % low
        # (profiles/r06d_pass_order.json: 13 070 behind the serialised pass's engine, 14 100 - 14 250 settled)
        if settle > 0:
            done_ = 0
            while done_ < settle // 3:
                k = min(total_steps, settle // 3 - done_)
                run(0, k, p32)
                done_ += k
            torch.cuda.synchronize()
        run(0, args.warmup, p32)
        torch.cuda.synchronize()
        e32, table32, n_sampled32 = timed_pass(p32, "exact-f32 pass")
        exact = {"value": round(D.whole_job_rate(n, args.steps, e32, world) / 2, 2),
                 "ms_per_step": round(1e3 * e32 / args.steps, 3), "dtype": "f32",
                 "note": "same job with precision='f32' (v_mfma_f32_16x16x4_f32 everywhere): the reference's own "
                         "arithmetic; per-kernel brackets and roofline collected exactly like the headline pass",
                 "host_step_period_ms": period.get("exact-f32 pass"),
                 "_table": table32, "_sampled": n_sampled32,
                 "_serial": serial_pass("f32") if args.serial_steps > 0 else None}

    if rank == 0:
        cps = D.whole_job_rate(n, args.steps, elapsed, world)
        pmc = pmc_live(precision, args) if (world == 1 and args.pmc != "off") else None
        pmc32 = pmc_live("f32", args) if (pmc is not None and exact is not None and args.pmc == "all") else None
        # per-kernel roofline from the SERIALISED pass (alone-times, reproducible from rocprofv3's kernel stats of
        # `--serial-only`); the brackets of the timed region — taken while `lanes` steps share the chip — go to the details
        # file as `roofline_kernels_overlapped` and are never quoted as kernel efficiency
        _, per_kernel_ovl = build_roofline(table, precision, n_sampled, pmc)
        if serial is not None:
            roof, per_kernel = build_roofline(serial[0], precision, serial[1], pmc)
            roof["serialised"], roof["serialised_ms_per_step"] = True, round(serial[2], 3)
        else:
            roof, per_kernel = build_roofline(table, precision, n_sampled, pmc)
            roof["serialised"] = False
        peak_path = PEAK_F32_MATRIX_TFLOPS if precision == "f32" else PEAK_F16_MATRIX_TFLOPS / SPLIT_PRODUCTS
        roof["whole_path_tflops"] = round(cps / world * ALG_GFLOP_PER_CHUNK / 1e3, 2)
        whole_path_frac = round(roof["whole_path_tflops"] / peak_path, 4)
        roof["exact_f32_value"] = exact["value"] if exact else None
        roof["host_fed_value"] = host_fed["value"] if host_fed else None
        whole_path_frac_f32 = None
        if exact is not None:
            ser32 = exact.pop("_serial")
            t32, s32 = exact.pop("_table"), exact.pop("_sampled")
            _, exact["roofline_kernels_overlapped"] = build_roofline(t32, "f32", s32, pmc32)
            r32, pk32 = build_roofline(ser32[0], "f32", ser32[1], pmc32) if ser32 else build_roofline(t32, "f32", s32, pmc32)
            r32["serialised"] = ser32 is not None
            if ser32:
                r32["serialised_ms_per_step"] = round(ser32[2], 3)
            r32["whole_path_tflops"] = round(2 * exact["value"] / world * ALG_GFLOP_PER_CHUNK / 1e3, 2)
            whole_path_frac_f32 = round(r32["whole_path_tflops"] / PEAK_F32_MATRIX_TFLOPS, 4)
            exact["roofline"], exact["roofline_kernels"] = r32, pk32
        # step-level utilisation: launches per step and PMC per-launch figures x the timed region's step time
        mfma_step, hbm_step, roof_mfma = step_level(per_kernel, 1e3 * elapsed / args.steps, precision, roof.get("traffic_source"))
        if exact is not None:
            exact["mfma_util_step"], exact["hbm_gbps_step"], exact["roofline_mfma"] = step_level(
                exact["roofline_kernels"], exact["ms_per_step"], "f32", exact["roofline"].get("traffic_source"))
        out = {
            "metric": "real-time-factor xRT streams/GPU @500ms step", "value": round(cps / 2, 2),
            "unit": "xRT 16 kHz streams (chunks/s / 2)", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if precision == "f32" else "f16x3",
            "dtype_note": ("exact-f32 MFMA (v_mfma_f32_16x16x4_f32), f32 VALU, f32 accumulation; clustering f64"
                           if precision == "f32" else
                           "every GEMM-shaped layer (sinc conv0, conv1/2, LSTM projections, MLP, TDNN 1-5): f32 operands "
                           "split into two f16 numbers (hi + lo*2^-11 = 22 mantissa bits), three f16 MFMAs per product, "
                           "f32 accumulation; operands beyond +-65504 are flagged as an error, not silently clamped "
                           "(dz_range_check).  LSTM recurrence, classifier, pooling, Linear(3000,512): exact f32; "
                           "clustering / aggregation f64.  Error vs the f32 oracle equals the exact-f32 path's "
                           "(tests/test_gpu_models.py, tests/test_gpu_parity_r2.py: same gates for both).  The "
                           "exact-f32 run of the same job is in `exact_f32`."),
            "data": "synthetic",
            "config": {"workload": "configs[1]: single MI355X, 5 s window / 500 ms step, "
                                   "pyannote/segmentation + pyannote/embedding architectures "
                                   "(random-init weights), %d concurrent synthetic 16 kHz streams per GPU" % n,
                       "streams_per_gpu": n, "chunks_per_step": world * n, "parallelism": f"streams x{world}",
                       "dist_backend": torch.distributed.get_backend() if world > 1 else None,
                       "rccl_ranks": (torch.distributed.get_world_size() if world > 1 and torch.distributed.get_backend() == "nccl"
                                      else 0), "cpu_affinity": affinity,
                       "weights_abs_sum_per_rank": wsums, "host_threads_per_rank": host_threads,
                       "steps_in_flight": pipe.max_inflight, "lanes": pipe.depth, "recurrence": pipe.recurrence or "valu", "seg_sub_batches": pipe.seg_split,
                       "settle_steps": settle, "engine": "StreamBatch(%s)" % ", ".join(
                           f"{k}={v}" for k, v in (("recurrence", args.recurrence), ("lanes", args.lanes), ("inflight", args.inflight)) if v) ,
                       "hip_streams": pipe.num_hip_streams,
                       "exact_f32_value": exact["value"] if exact else None,
                       "host_fed_value": host_fed["value"] if host_fed else None},
            "roofline": roof, "roofline_mfma": roof_mfma, "mfma_util_step": mfma_step, "hbm_gbps_step": hbm_step,
            "whole_path_frac": whole_path_frac, "whole_path_frac_exact_f32": whole_path_frac_f32, "power": power,
            "roofline_kernels": per_kernel, "roofline_kernels_overlapped": per_kernel_ovl,
            "roofline_sampling": (f"`roofline*`: the serialised pass ({serial[1]} steps on one lane / one HIP stream, every launch "
                                  f"bracketed, {serial[2]:.3f} ms per step); " if serial else "") +
                                 (f"`roofline_kernels_overlapped`: {n_sampled} of the {args.steps} timed steps (every {PROF_EVERY}th) "
                                  "carried the per-kernel event pairs while the lanes overlap" if n_sampled else
                                  "the timed passes carry no event pairs (--overlap-brackets takes them on every 10th step)"),
            "exact_f32": exact,
            "host_fed": host_fed,
            "host": host_line,
        }
        if world == 1 and not args.no_rehearsal:
            out["host_rehearsal"] = host_rehearsal(args, precision, usable)
        if args.kernel_table:
            Path(args.kernel_table).parent.mkdir(parents=True, exist_ok=True)
            Path(args.kernel_table).write_text(json.dumps(table, indent=1))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_chunks)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        emit(out, args.details)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
