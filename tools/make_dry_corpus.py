#!/usr/bin/env python
"""A stand-in corpus for a DRY RUN of tools/verify_real.py on a box without the gated weights / AMI audio:
synthetic checkpoints in the layout of the real ones (a Lightning-style `segmentation.ckpt` with the `model.` prefix,
a plain `embedding.bin`) and N synthetic 16 kHz WAV files.  usage: make_dry_corpus.py OUT_DIR [files] [seconds]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from diart_amd.inference import write_wav  # noqa: E402
from diart_amd.synth import synth_embedding_state, synth_segmentation_state, synth_stream  # noqa: E402

out = Path(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 120.0
(out / "ckpt").mkdir(parents=True, exist_ok=True)
(out / "ami").mkdir(parents=True, exist_ok=True)
torch.save({"state_dict": {"model." + k: v for k, v in synth_segmentation_state().items()}, "epoch": 0},
           out / "ckpt" / "segmentation.ckpt")
torch.save(dict(synth_embedding_state()), out / "ckpt" / "embedding.bin")
for i in range(n):
    write_wav(out / "ami" / f"DRY{i:02d}.wav", synth_stream(2000 + i, seconds * 0.9 ** i), 16000)
print(out)
