"""The subset of lagrangebench/defaults.py (:7-176) the rollout path reads, as plain dicts
(OmegaConf is not a dependency; any Mapping with the same keys is accepted and merged onto
these defaults exactly as case.py:91-98 / rollout.py:345-349 do)."""
from __future__ import annotations

import copy
from typing import Any, Mapping


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

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
    "train": {                                # defaults.py:66-109
        "batch_size": 1, "step_max": 500_000, "num_workers": 4, "noise_std": 3e-4,
        "optimizer": {"lr_start": 1e-4, "lr_final": 1e-6, "lr_decay_rate": 0.1, "lr_decay_steps": 1e5},
        "pushforward": {"steps": [-1, 20000, 300000, 400000], "unrolls": [0, 1, 2, 3], "probs": [18, 2, 1, 1]},
        "loss_weight": {"acc": 1.0, "vel": 0.0, "pos": 0.0},
    },
    "logging": {"log_steps": 1000, "eval_steps": 10000, "wandb": False, "wandb_project": None,
                "wandb_entity": "lagrangebench", "ckp_dir": "ckp", "run_name": None},  # defaults.py:153-168
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
