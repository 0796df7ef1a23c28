"""CPU oracle (TEST INFRASTRUCTURE ONLY) of the `sinkhorn` rollout metric.

Reference: lagrangebench/evaluate/metrics.py
  * :127-136  per strided frame pair (pred[0::stride], target[0::stride]) -> self.sinkhorn(pred, target)
  * :162-176  _sinkhorn_ott: cost matrices xy / xx / yy from _distance_matrix, uniform weights,
              ott.tools.sinkhorn_divergence.sinkhorn_divergence(Geometry, C_xy, C_xx, C_yy, a=, b=,
              sinkhorn_kwargs={"threshold": 1e-4}).divergence
  * :198-213  _distance_matrix: C[i, j] = sum(displacement_fn(x_i, y_j)**2), cast to float32

The optimisation itself lives in the third-party ott-jax (pinned 0.4.6, poetry.lock:2009-2010), which
is NOT under /root/reference and cannot be installed here -> restated from its published algorithm
([mem] = recollection of ott 0.4.x sources, to be re-checked by tests/golden/make_jax_golden.py
when a JAX machine is available; "parity unpinned" until then):
  [mem] Geometry(cost_matrix) with epsilon=None: epsilon = 0.05 * mean(cost_matrix); in
        sinkhorn_divergence (share_epsilon=True) the xx and yy problems reuse the xy epsilon.
  [mem] Sinkhorn defaults: lse_mode, zero-initialised dual potentials, inner_iterations=10 (the
        error is evaluated every 10th iteration), max_iterations=2000, norm_error=1, momentum 1.0,
        sequential updates: g <- eps*log b - eps*LSE_i((f_i - C_ij)/eps), then f with the NEW g.
        Error (balanced, sequential) = || marginal over i of P - b ||_1, P_ij = exp((f_i+g_j-C_ij)/eps).
  [mem] symmetric terms (xx, yy): parallel_dual_updates=True with momentum 0.5 (both potentials
        updated from the OLD pair and averaged with their old value); the error is then the sum of
        both marginal errors.
  [mem] reg_ot_cost = sum_i a_i (f_i - eps log a_i) + sum_j b_j (g_j - eps log b_j)
                      + eps * (sum(a) sum(b) - sum(P));
        divergence = reg_xy - 0.5 * (reg_xx + reg_yy) + 0.5 * eps * (sum(a) - sum(b))**2.
At convergence this is the standard debiased Sinkhorn divergence S_eps(x, y); `test_sinkhorn_*`
pin the solver against the closed form of the 2 x 2 entropic problem.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np


def distance_matrix(displacement_fn: Callable, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """metrics.py:198-213 (squared=True): C[i, j] = |disp(x_i, y_j)|^2 as float32."""
    d = displacement_fn(x[:, None, :], y[None, :, :])
    return (d ** 2).sum(-1).astype(np.float32)


def _lse(z: np.ndarray, axis: int) -> np.ndarray:
    m = z.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(z - m).sum(axis=axis, keepdims=True))).squeeze(axis)


def sinkhorn_solve(C: np.ndarray, a: np.ndarray, b: np.ndarray, eps: float, threshold: float = 1e-4,
                   parallel: bool = False, momentum: float = 1.0, inner_iterations: int = 10,
                   max_iterations: int = 2000) -> Tuple[float, int, float]:
    """-> (reg_ot_cost, iterations run, last error).  C float32 (n, m); everything else float64."""
    C = C.astype(np.float64)
    f, g = np.zeros(len(a)), np.zeros(len(b))
    la, lb = np.log(a), np.log(b)
    err, it = np.inf, 0

    def marginals(f, g):
        P = np.exp((f[:, None] + g[None, :] - C) / eps)
        return P.sum(axis=1), P.sum(axis=0)

    while it < max_iterations:
        g_new = eps * lb - eps * _lse((f[:, None] - C) / eps, axis=0)
        g_upd = (1.0 - momentum) * g + momentum * g_new
        g_for_f = g if parallel else g_upd
        f_new = eps * la - eps * _lse((g_for_f[None, :] - C) / eps, axis=1)
        f = (1.0 - momentum) * f + momentum * f_new
        g = g_upd
        it += 1
        if it % inner_iterations == 0:
            ma, mb = marginals(f, g)
            err = np.abs(mb - b).sum()
            if parallel:
                err += np.abs(ma - a).sum()
            if err < threshold:
                break
    ma, mb = marginals(f, g)
    reg = float((a * (f - eps * la)).sum() + (b * (g - eps * lb)).sum() + eps * (a.sum() * b.sum() - mb.sum()))
    return reg, it, float(err)


def sinkhorn_divergence(displacement_fn: Callable, pred: np.ndarray, target: np.ndarray,
                        threshold: float = 1e-4, return_info: bool = False):
    """metrics.py:162-176 for one frame pair (N, dim) x (M, dim)."""
    Cxy = distance_matrix(displacement_fn, pred, target)
    Cxx = distance_matrix(displacement_fn, pred, pred)
    Cyy = distance_matrix(displacement_fn, target, target)
    a = np.ones(len(pred)) / len(pred)
    b = np.ones(len(target)) / len(target)
    eps = 0.05 * float(Cxy.astype(np.float64).mean())
    rxy, ixy, exy = sinkhorn_solve(Cxy, a, b, eps, threshold)
    rxx, ixx, exx = sinkhorn_solve(Cxx, a, a, eps, threshold, parallel=True, momentum=0.5)
    ryy, iyy, eyy = sinkhorn_solve(Cyy, b, b, eps, threshold, parallel=True, momentum=0.5)
    div = rxy - 0.5 * (rxx + ryy) + 0.5 * eps * (a.sum() - b.sum()) ** 2
    if return_info:
        return div, {"eps": eps, "iters": (ixy, ixx, iyy), "errors": (exy, exx, eyy), "reg": (rxy, rxx, ryy)}
    return div


def sinkhorn_rollout(displacement_fn: Callable, pred_rollout: np.ndarray, target_rollout: np.ndarray,
                     stride: int) -> np.ndarray:
    """metrics.py:127-136: one divergence per strided frame of (T, N, dim) rollouts."""
    return np.array([sinkhorn_divergence(displacement_fn, p, t)
                     for p, t in zip(pred_rollout[0::stride], target_rollout[0::stride])])
