"""diart's per-chunk diarization hot path on MI355X (gfx950): hand-written HIP kernels behind
diart's own operator API (SegmentationModel, EmbeddingModel, OnlineSpeakerClustering)."""
__version__ = "0.1.0"
