from .aggregation import (AggregationStrategy, AverageStrategy, BatchedOutputTail, DelayedAggregation,
                          FirstOnlyStrategy, HammingWeightedAverageStrategy)
from .base import HyperParameter, Pipeline, PipelineConfig
from .clustering import (BatchedSpeakerClustering, IncrementalSpeakerClustering,
                         OnlineSpeakerClustering)
from .diarization import SpeakerDiarization, SpeakerDiarizationConfig
from .embedding import (EmbeddingNormalization, OverlapAwareSpeakerEmbedding,
                        OverlappedSpeechPenalty, SpeakerEmbedding)
from .segmentation import SpeakerSegmentation
from .utils import Binarize
from .vad import VoiceActivityDetection, VoiceActivityDetectionConfig

__all__ = ["SpeakerSegmentation", "SpeakerEmbedding", "OverlappedSpeechPenalty",
           "EmbeddingNormalization", "OverlapAwareSpeakerEmbedding", "OnlineSpeakerClustering",
           "IncrementalSpeakerClustering", "BatchedSpeakerClustering", "DelayedAggregation",
           "AggregationStrategy", "HammingWeightedAverageStrategy", "AverageStrategy", "FirstOnlyStrategy",
           "BatchedOutputTail", "Binarize", "Pipeline", "PipelineConfig", "HyperParameter",
           "SpeakerDiarization", "SpeakerDiarizationConfig", "VoiceActivityDetection",
           "VoiceActivityDetectionConfig"]
