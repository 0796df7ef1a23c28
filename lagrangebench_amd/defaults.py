"""The subset of lagrangebench/defaults.py (:7-176) the rollout path reads, as plain dicts
(OmegaConf is not a dependency; any Mapping with the same keys is accepted and merged onto
these defaults exactly as case.py:91-98 / rollout.py:345-349 do)."""
from __future__ import annotations

import copy
from typing import Any, Mapping


class _AttrDict(dict):
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(d):
    if isinstance(d, Mapping):
        return _AttrDict((k, _wrap(v)) for k, v in d.items())
    return d


defaults = _wrap({
    "seed": 0,
    "dtype": "float64",                       # defaults.py:22
    "model": {
        "name": None, "input_seq_length": 6, "num_mp_steps": 10, "num_mlp_layers": 2,
        "latent_dim": 128, "isotropic_norm": False, "magnitude_features": False,
        "lmax_attributes": 1, "lmax_hidden": 1, "segnn_norm": "none", "velocity_aggregate": "avg",
    },                                        # defaults.py:38-63
    "train": {"noise_std": 3e-4},             # defaults.py:75
    "eval": {
        "n_rollout_steps": 20,                # defaults.py:113
        "rollout_dir": None,
        "train": {"n_trajs": 50, "metrics_stride": 10, "batch_size": 1, "metrics": ["mse"],
                  "out_type": "none"},                       # defaults.py:121-134
        "infer": {"n_trajs": -1, "metrics_stride": 1, "batch_size": 2, "metrics": ["mse", "e_kin", "sinkhorn"],
                  "out_type": "pkl", "n_extrap_steps": 0},   # defaults.py:136-150
    },
    "neighbors": {"backend": "jaxmd_vmap", "multiplier": 1.25},  # defaults.py:170-175
})


def merge(base: Mapping, override: Any):
    """OmegaConf.merge(defaults.x, cfg_x) for plain mappings (None -> base)."""
    out = copy.deepcopy(dict(base))
    if override is None:
        return _wrap(out)
    for k, v in dict(override).items():
        if isinstance(v, Mapping) and isinstance(out.get(k), Mapping):
            out[k] = merge(out[k], v)
        else:
            out[k] = v
    return _wrap(out)
