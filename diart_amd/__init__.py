"""diart's per-chunk diarization hot path on MI355X (gfx950): hand-written HIP kernels behind
diart's own operator API (SegmentationModel, EmbeddingModel, OnlineSpeakerClustering)."""
__version__ = "0.1.0"

import os as _os

# StreamBatch keeps several HIP streams busy at once (segmentation lanes + the embedding stream +
# the caller's).  The ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4) and streams that share a queue run one after the other: with the default the lanes
# serialised (measured: 1.74 -> 1.38 ms per 64-stream step with 8 queues; 12+ made the host-side
# launches slower again).  The variable is read when the HIP runtime initialises, so it has to be
# in the environment before the first GPU call of the process.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
