"""CPU restatement of the reference's small tensor functions (TEST INFRASTRUCTURE).

Pinned: ``tests/golden/make_golden.py`` runs the reference's own
``/root/reference/src/diart/functional.py`` and ``blocks/embedding.py`` on seeded inputs and
``tests/test_oracle_golden.py`` compares these restatements with the committed outputs.
"""
from __future__ import annotations

import torch


def overlapped_speech_penalty_ref(segmentation: torch.Tensor, gamma: float = 3, beta: float = 10,
                                  normalize: bool = False) -> torch.Tensor:
    """functional.py:6-13 (+ the min-max option of blocks/embedding.py:102-106).

    segmentation (batch, frames, speakers) -> weights of the same shape.
    """
    probs = torch.softmax(beta * segmentation, dim=-1)
    weights = torch.pow(segmentation, gamma) * torch.pow(probs, gamma)
    weights = torch.where(weights < 1e-8, torch.full_like(weights, 1e-8), weights)
    if normalize:
        lo = weights.min(dim=1, keepdim=True).values
        hi = weights.max(dim=1, keepdim=True).values
        weights = torch.nan_to_num((weights - lo) / (hi - lo), nan=1e-8)
    return weights


def normalize_embeddings_ref(embeddings: torch.Tensor, norm: float | torch.Tensor = 1) -> torch.Tensor:
    """functional.py:16-27: (batch, speakers, feat) or (speakers, feat) -> 3-D, L2 norm = norm."""
    if embeddings.ndim == 2:
        embeddings = embeddings.unsqueeze(0)
    return norm * embeddings / torch.norm(embeddings, p=2, dim=-1, keepdim=True)
