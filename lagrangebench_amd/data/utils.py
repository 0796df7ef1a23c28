"""Data utils mirrored from lagrangebench/data/utils.py."""
from __future__ import annotations

from typing import Dict, List

import numpy as np


def get_dataset_stats(metadata: Dict[str, List[float]], is_isotropic_norm: bool,
                      noise_std: float, dtype=np.float64) -> Dict[str, Dict[str, np.ndarray]]:
    """lagrangebench/data/utils.py:9-45 (host-side, a handful of scalars in the case's dtype)."""
    acc_mean = np.array(metadata["acc_mean"], dtype=dtype)
    acc_std = np.array(metadata["acc_std"], dtype=dtype)
    vel_mean = np.array(metadata["vel_mean"], dtype=dtype)
    vel_std = np.array(metadata["vel_std"], dtype=dtype)
    if is_isotropic_norm:
        acc_mean = np.mean(acc_mean) * np.ones_like(acc_mean)
        acc_std = np.sqrt(np.mean(acc_std**2)) * np.ones_like(acc_std)
        vel_mean = np.mean(vel_mean) * np.ones_like(vel_mean)
        vel_std = np.sqrt(np.mean(vel_std**2)) * np.ones_like(vel_std)
    return {
        "acceleration": {"mean": acc_mean, "std": np.sqrt(acc_std**2 + noise_std**2)},
        "velocity": {"mean": vel_mean, "std": np.sqrt(vel_std**2 + noise_std**2)},
    }


def numpy_collate(batch) -> np.ndarray:
    """lagrangebench/data/utils.py:48-56."""
    if isinstance(batch[0], np.ndarray):
        return np.stack(batch)
    if isinstance(batch[0], (tuple, list)):
        return type(batch[0])(numpy_collate(samples) for samples in zip(*batch))
    return np.asarray(batch)
