"""Rollout metrics - mirror of lagrangebench/evaluate/metrics.py.

``mse`` and ``mae`` (metrics.py:86-96,139-147), ``e_kin`` (metrics.py:98-125,157-160) and
``sinkhorn`` (metrics.py:127-136,162-176, the default OTT backend) are computed on the device by
lb_metrics / lb_ekin / lb_sinkhorn.  ``ot_backend="pot"`` (metrics.py:178-196: POT's sinkhorn2 with a
fixed reg=0.1, which the reference runs through a host callback) is lb_sinkhorn_pot, also on the device.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

MetricsDict = Dict[str, Dict[str, torch.Tensor]]


class MetricsComputer:
    METRICS = ["mse", "mae", "sinkhorn", "e_kin"]

    def __init__(self, active_metrics: List, dist_fn: Callable, metadata: Dict, input_seq_length: int,
                 stride: int = 10, loss_ranges: Optional[List] = None, ot_backend: str = "ott",
                 case=None):
        """Same arguments as the reference (metrics.py:30-67) plus ``case``: the CaseSetupFn whose
        engine evaluates the metric kernels (``dist_fn`` is kept for signature parity)."""
        if active_metrics is None:
            active_metrics = []
        assert all(m in self.METRICS for m in active_metrics)
        assert ot_backend in ["ott", "pot"]
        self.ot_backend = ot_backend
        self._active_metrics = list(active_metrics)
        self._dist_fn = dist_fn
        self._loss_ranges = loss_ranges if loss_ranges is not None else [1, 5, 10, 20, 50, 100]
        self._input_seq_length = input_seq_length
        self._stride = stride
        self._metadata = metadata
        self._case = case

    def __call__(self, pred_rollout, target_rollout) -> Dict[str, torch.Tensor]:
        """pred/target: (T, N, dim) or batched (B, T, N, dim) - the reference vmaps over B
        (rollout.py:228)."""
        if self._case is None:
            raise RuntimeError("MetricsComputer needs case=<CaseSetupFn> to reach the HIP engine")
        pred = pred_rollout if isinstance(pred_rollout, torch.Tensor) else torch.as_tensor(np.asarray(pred_rollout))
        tgt = target_rollout if isinstance(target_rollout, torch.Tensor) else torch.as_tensor(np.asarray(target_rollout))
        batched = pred.dim() == 4
        if not batched:
            pred, tgt = pred[None], tgt[None]
        T = pred.shape[1]
        eng = self._case.engine(pred.shape[0])
        want = [m for m in self._active_metrics if m in ("mse", "mae")]
        res = eng.metrics(pred, tgt[:, :T], T, want=want) if want else {}
        out: Dict[str, torch.Tensor] = {}
        if "e_kin" in self._active_metrics:
            # metrics.py:98-125: strided velocities -> kinetic energy, predicted vs target, and its MSE
            dt = self._metadata["dt"] * self._metadata["write_every"]
            dx = self._metadata["dx"]
            ek_p = eng.ekin(pred, self._stride, dt, dx)
            ek_t = eng.ekin(tgt[:, :T], self._stride, dt, dx)
            mse_e = ((ek_p - ek_t) ** 2).mean(dim=1)
            if batched:
                out["e_kin"] = {"predicted": ek_p, "target": ek_t, "mse": mse_e}
            else:
                out["e_kin"] = {"predicted": ek_p[0], "target": ek_t[0], "mse": mse_e[0]}
        if "sinkhorn" in self._active_metrics:
            # metrics.py:127-136: one divergence per stride-th frame pair
            if self.ot_backend == "ott":
                sk = eng.sinkhorn(pred, tgt[:, :T], self._stride)
            else:  # metrics.py:178-196: POT sinkhorn2(reg=0.1, numItermax=500, stopThr=1e-05), float32 result
                sk = eng.sinkhorn_pot(pred, tgt[:, :T], self._stride).to(torch.float32)
            out["sinkhorn"] = sk if batched else sk[0]
        for name in want:
            v = res[name] if batched else res[name][0]
            out[name] = v
            for i in self._loss_ranges:
                if i < T:  # only horizons strictly shorter than the rollout (metrics.py:94-96)
                    out[f"{name}{i}"] = v[..., :i]
        return out


def averaged_metrics(eval_metrics: MetricsDict) -> Dict[str, float]:
    """metrics.py:233-252: per rollout the mean of every metric is appended to a list per key
    (``mse``/``mae`` both land in ``loss``, ``e_kin`` contributes its ``mse``); the result holds the
    mean ``val/{k}`` and the standard deviation ``val/std{k}`` of each list."""
    trajectory_averages = defaultdict(list)
    for rollout in eval_metrics.values():
        for k, v in rollout.items():
            if k == "e_kin":
                v = v["mse"]
            if k in ["mse", "mae"]:
                k = "loss"
            trajectory_averages[k].append(float(torch.as_tensor(v).double().mean()))
    small_metrics = {}
    for k, v in trajectory_averages.items():
        small_metrics[f"val/{k}"] = float(np.mean(v))
    for k, v in trajectory_averages.items():
        small_metrics[f"val/std{k}"] = float(np.std(v))
    return small_metrics
