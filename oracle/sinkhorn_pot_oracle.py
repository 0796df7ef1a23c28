"""CPU oracle (TEST INFRASTRUCTURE ONLY) of the `sinkhorn` rollout metric with ``ot_backend="pot"``.

Reference: lagrangebench/evaluate/metrics.py
  * :178-187  _sinkhorn_pot: clip(S(pred, target) - 0.5 * (S(pred, pred) + S(target, target)), 0) as float32
  * :189-196  _custom_empirical_sinkorn_pot: uniform a, b; M = _distance_matrix (float32);
              float32(ot.bregman.sinkhorn2(a, b, M, reg=0.1, numItermax=500, stopThr=1e-05))
  * :198-213  _distance_matrix

The solver is third-party: POT ("Python Optimal Transport"), imported lazily by the reference and NOT listed in
its poetry.lock (no pinned version), not under /root/reference and not installable here -> "parity unpinned".
Restated from POT's published algorithm for method="sinkhorn" (ot.bregman.sinkhorn_knopp, Cuturi 2013):
  [mem] u = ones(n)/n, v = ones(m)/m; K = exp(M / (-reg)); Kp = (1/a)[:, None] * K
  [mem] for ii in range(numItermax):
            KtU = K.T @ u;  v = b / KtU;  u = 1 / (Kp @ v)
            if any(KtU == 0) or any non-finite entry of u or v: restore the previous (u, v) and stop
            if ii % 10 == 0:  err = || u @ (K * v) - b ||_2 (einsum 'i,ij,j->j');  stop if err < stopThr
  [mem] sinkhorn2 returns sum(u[:, None] * K * v[None, :] * M).
POT computes in the dtype of its inputs (float32 here); this restatement and the HIP path use float64 on the
float32 cost matrix, so the two agree with each other to rounding and with POT to float32 accuracy of the
iteration (the test tolerance says which).
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np

from .sinkhorn_oracle import distance_matrix


def sinkhorn2(a: np.ndarray, b: np.ndarray, M: np.ndarray, reg: float = 0.1, numItermax: int = 500,
              stopThr: float = 1e-5) -> Tuple[float, int, int]:
    """-> (value, iterations run, how the loop ended: 0 numItermax / 1 converged / 2 numerical stop)."""
    M = M.astype(np.float64)
    u, v = np.ones(len(a)) / len(a), np.ones(len(b)) / len(b)
    K = np.exp(M / (-reg))
    Kp = (1.0 / a)[:, None] * K
    how, it = 0, 0
    for ii in range(numItermax):
        uprev, vprev = u, v
        KtU = K.T @ u
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            v = b / KtU
            u = 1.0 / (Kp @ v)
        it = ii + 1
        if (KtU == 0).any() or not np.isfinite(u).all() or not np.isfinite(v).all():
            u, v, how = uprev, vprev, 2
            break
        if ii % 10 == 0:
            err = np.linalg.norm(np.einsum("i,ij,j->j", u, K, v) - b)
            if err < stopThr:
                how = 1
                break
    return float((u[:, None] * K * v[None, :] * M).sum()), it, how


def sinkhorn_divergence_pot(displacement_fn: Callable, pred: np.ndarray, target: np.ndarray, return_info=False):
    """metrics.py:178-196 for one frame pair; float32 arithmetic of the final combination as in the reference."""
    a = np.ones(len(pred)) / len(pred)
    b = np.ones(len(target)) / len(target)
    res = [sinkhorn2(p, q, distance_matrix(displacement_fn, x, y))
           for (p, q, x, y) in ((a, b, pred, target), (a, a, pred, pred), (b, b, target, target))]
    ab, aa, bb = (np.float32(r[0]) for r in res)
    d = np.float32(ab - np.float32(0.5) * (aa + bb))
    out = np.float32(np.clip(d, 0, None))
    if return_info:
        return out, {"values": (ab, aa, bb), "iters": tuple(r[1] for r in res), "how": tuple(r[2] for r in res)}
    return out
