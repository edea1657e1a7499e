from .clustering import (BatchedSpeakerClustering, IncrementalSpeakerClustering,
                         OnlineSpeakerClustering)
from .embedding import (EmbeddingNormalization, OverlapAwareSpeakerEmbedding,
                        OverlappedSpeechPenalty, SpeakerEmbedding)
from .segmentation import SpeakerSegmentation

__all__ = ["SpeakerSegmentation", "SpeakerEmbedding", "OverlappedSpeechPenalty",
           "EmbeddingNormalization", "OverlapAwareSpeakerEmbedding", "OnlineSpeakerClustering",
           "IncrementalSpeakerClustering", "BatchedSpeakerClustering"]
