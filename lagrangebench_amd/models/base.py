"""BaseModel contract - lagrangebench/models/base.py:11-41."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict, Tuple


class BaseModel(ABC):
    """``model.init(key, sample) -> (params, state)`` and
    ``model.apply(params, state, sample) -> (pred_dict, state)`` are what the reference gets from
    ``hk.without_apply_rng(hk.transform_with_state(model))`` (runner.py:67); models here expose
    them directly.  ``sample = (features, particle_type)``; the prediction dict has one of the
    keys "acc" / "vel" / "pos" with a (N, dim) (or (B, N, dim)) array."""

    @abstractmethod
    def init(self, key, sample) -> Tuple[Dict, Dict]:
        raise NotImplementedError

    @abstractmethod
    def apply(self, params, state, sample) -> Tuple[Dict, Dict]:
        raise NotImplementedError
