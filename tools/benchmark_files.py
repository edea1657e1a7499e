#!/usr/bin/env python
"""BASELINE.json configs 1 / 4: the file-parallel evaluation (`diart.benchmark`) on MI355X.

    python tools/benchmark_files.py --files 1 --seconds 30                      # config 1 shape
    python tools/benchmark_files.py --gpus 8 --files 16 --seconds 600           # config 4 shape (starts its 8 ranks)

AMI-SDM audio and the gated pyannote checkpoints are not available offline, so the corpus is
synthetic (same generator as bench.py, written as 16-bit WAV) and the weights are the seeded
random ones; the run therefore reports throughput and writes RTTMs — accuracy parity of exactly
this path against the all-CPU chain is gated in tests/test_gpu_der.py.  One process per GPU,
whole files assigned by longest-processing-time, weights broadcast once over RCCL, one JSON line
from rank 0."""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diart_amd import distributed as D  # noqa: E402
from diart_amd import models as M  # noqa: E402
from diart_amd.blocks import SpeakerDiarization, SpeakerDiarizationConfig  # noqa: E402
from diart_amd.hostinfo import limit_host_threads  # noqa: E402
from diart_amd.inference import Benchmark, DistributedBenchmark, wav_duration, write_wav  # noqa: E402
from diart_amd.synth import (synth_ecapa_state, synth_embedding_state, synth_segmentation_state,  # noqa: E402
                             synth_stream)

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=0,
                help="start this many ranks (one per GPU, torch.distributed.run) when not already under torchrun")
ap.add_argument("--files", type=int, default=1)
ap.add_argument("--seconds", type=float, default=30.0, help="duration of the first file; file i is 7 %% shorter than i-1")
ap.add_argument("--batch-size", type=int, default=32)
ap.add_argument("--latency", type=float, default=0.5)
ap.add_argument("--workdir", type=str, default="gpurun_out/benchmark_files")
ap.add_argument("--ami-hparams", action="store_true", help="tau/rho/delta of README.md:391")
ap.add_argument("--config3", action="store_true",
                help="BASELINE config 3: powerset segmentation-3.0 + speechbrain ECAPA-TDNN embedding, "
                     "normalised OSP weights")
args = ap.parse_args()

rc = D.self_launch(args.gpus, str(Path(__file__).resolve()), sys.argv[1:])
if rc is not None:
    raise SystemExit(rc)
limit_host_threads()
rank, world, local = D.init_from_env()
if args.gpus and world != args.gpus:
    raise SystemExit(f"benchmark_files.py: WORLD_SIZE={world} but --gpus {args.gpus}")
import os  # noqa: E402
device = torch.device("cuda", int(os.environ.get("DZ_FORCE_DEVICE", local)))   # see bench.py
torch.cuda.set_device(device)
work = Path(args.workdir)
speech, out = work / "wav", work / f"rttm_w{world}"
if rank == 0:
    speech.mkdir(parents=True, exist_ok=True)
    for i in range(args.files):
        p = speech / f"synthetic_{i:02d}.wav"
        if not p.exists():
            write_wav(p, synth_stream(1000 + i, max(6.0, args.seconds * 0.93 ** i)), 16000)
if world > 1:
    torch.distributed.barrier()

make_seg = (lambda: synth_segmentation_state(seed=77, powerset=True)) if args.config3 else synth_segmentation_state
make_emb = synth_ecapa_state if args.config3 else synth_embedding_state
seg_sd = make_seg() if rank == 0 else None
emb_sd = make_emb() if rank == 0 else None
if world > 1:
    seg_sd = D.broadcast_state(seg_sd, D.state_spec(make_seg()), device)
    emb_sd = D.broadcast_state(emb_sd, D.state_spec(make_emb()), device)
hp = dict(tau_active=0.507, rho_update=0.006, delta_new=1.057) if args.ami_hparams else {}
if args.config3:
    hp.update(normalize_embedding_weights=True, tau_active=0.5)
cfg = SpeakerDiarizationConfig(
    segmentation=M.SegmentationModel.from_state(seg_sd, max_batch=args.batch_size, powerset=args.config3),
    embedding=M.EmbeddingModel.from_state(emb_sd, max_batch=(3 if args.config3 else 1) * args.batch_size),
    latency=args.latency, device=device, **hp)
bench = DistributedBenchmark(Benchmark(speech, None, out, show_report=False, batch_size=args.batch_size))
# warm-up: weights packed, arenas allocated, kernels loaded
SpeakerDiarization(cfg)
cfg.segmentation(torch.zeros(1, 1, 80000, device=device))
fb = bench.benchmark.file_batch(SpeakerDiarization, cfg)
if fb is not None:      # the batched engine's arenas / pinned slots, like the models' above
    import numpy as np  # noqa: E402
    fb.run([("warm-up", np.zeros(16000 * 12, dtype=np.float32), 0.0)])
torch.cuda.synchronize()
if world > 1:
    torch.distributed.barrier()
t0 = time.perf_counter()
uris = bench(SpeakerDiarization, cfg)
torch.cuda.synchronize()
if world > 1:
    torch.distributed.barrier()
dt = time.perf_counter() - t0
if rank == 0:
    audio_s = sum(wav_duration(p) for p in sorted(speech.glob("*.wav"))[:args.files])
    chunks = sum(max(0, int((wav_duration(p) + args.latency - 0.5 - 5.0) / 0.5) + 1)
                 for p in sorted(speech.glob("*.wav"))[:args.files])
    print(json.dumps({"models": "segmentation-3.0 (powerset) + ECAPA-TDNN" if args.config3 else
                                "pyannote/segmentation + pyannote/embedding",
                      "workload": f"{args.files} synthetic 16 kHz files, {audio_s:.0f} s of audio, batch "
                                  f"{args.batch_size}, latency {args.latency}", "n_gpus": world, "files": len(uris),
                      "path": bench.benchmark.last_path, "wall_s": round(dt, 3), "audio_seconds_per_second": round(audio_s / dt, 1),
                      "chunks_per_second": round(chunks / dt, 1), "rttm_dir": str(out)}), flush=True)
if world > 1:
    torch.distributed.destroy_process_group()
