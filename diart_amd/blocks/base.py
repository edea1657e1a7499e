"""Pipeline / PipelineConfig interfaces and tunable hyper-parameters (reference:
``/root/reference/src/diart/blocks/base.py:12-137``) — what ``StreamingInference``,
``Benchmark`` and ``Optimizer`` rely on."""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Any, Sequence, Tuple

from ..features import SlidingWindowFeature


@dataclass
class HyperParameter:
    name: str
    low: float
    high: float

    @staticmethod
    def from_name(name: str) -> "HyperParameter":
        for hp in (TauActive, RhoUpdate, DeltaNew):
            if hp.name == name:
                return hp
        raise ValueError(f"Hyper-parameter '{name}' not recognized")


TauActive = HyperParameter("tau_active", low=0, high=1)
RhoUpdate = HyperParameter("rho_update", low=0, high=1)
DeltaNew = HyperParameter("delta_new", low=0, high=2)


class PipelineConfig(ABC):
    @property
    @abstractmethod
    def duration(self) -> float: ...

    @property
    @abstractmethod
    def step(self) -> float: ...

    @property
    @abstractmethod
    def latency(self) -> float: ...

    @property
    @abstractmethod
    def sample_rate(self) -> int: ...

    def get_padding(self, stream_duration: float) -> Tuple[float, float]:
        """(left, right) zero padding in seconds for a stream of this length — the arithmetic of
        ``get_file_padding`` (base.py:81-85, utils.py:69-88) without the audio-file probe."""
        right = self.latency - self.step
        total = stream_duration + right
        left = self.duration - total if total < self.duration else 0
        return left, right

    def get_file_padding(self, filepath) -> Tuple[float, float]:
        """The reference's entry point (``blocks/base.py:81-85`` -> ``utils.get_padding_left/right``):
        padding for an audio FILE; the duration comes from the WAV header."""
        from ..inference import wav_duration
        return self.get_padding(wav_duration(filepath))


class Pipeline(ABC):
    @staticmethod
    @abstractmethod
    def get_config_class() -> type: ...

    @staticmethod
    @abstractmethod
    def suggest_metric(): ...

    @staticmethod
    @abstractmethod
    def hyper_parameters() -> Sequence[HyperParameter]: ...

    @property
    @abstractmethod
    def config(self) -> PipelineConfig: ...

    @abstractmethod
    def reset(self): ...

    @abstractmethod
    def set_timestamp_shift(self, shift: float): ...

    @abstractmethod
    def __call__(self, waveforms: Sequence[SlidingWindowFeature]) -> Sequence[Tuple[Any, SlidingWindowFeature]]: ...
