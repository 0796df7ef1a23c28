"""Differentiable SEGNN forward in torch (lagrangebench/models/segnn.py:44-362,595-610 through e3nn-jax): TEST INFRASTRUCTURE -
the checker of the device SEGNN training step (csrc/lb_train_segnn.h: lb_segnn_train_loss_grad).  It restates
oracle/segnn_oracle.py (same assumptions A1-A6, same **parity unpinned** caveat: e3nn-jax cannot be executed here) operation
for operation on torch tensors, so that torch.autograd gives reference gradients of trainer.py:35-60's _mse with respect to every
tensor-product weight; `tests/test_segnn_train.py` checks this file's forward against the NumPy oracle and the engine's
gradients against it.  The geometric inputs (node features, attributes, message features: SEGNN._transform, segnn.py:513-587)
carry no gradient - they are taken from oracle.segnn_oracle.segnn_transform.  Only tests import this file.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from oracle import segnn_oracle as S

SV = Tuple[torch.Tensor, torch.Tensor]  # scalars (R, ns), vectors (R, nv, 3)


def params_to_torch(params, dtype=torch.float64, requires_grad: bool = False) -> Dict[str, Dict[str, torch.Tensor]]:
    out = {}
    for name, blk in params.items():
        if isinstance(blk, dict):
            out[name] = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad) for k, v in blk.items()}
    return out


def _silu(x):
    return x * torch.sigmoid(x)


def tp_inputs(ops: List[SV], attr: torch.Tensor):
    """segnn_oracle.tp_inputs: scalar channels (R, K), vector channels (R, K, 3)."""
    a0, a = attr[:, :1], attr[:, 1:4]
    xs, xv = [], []
    for s, v in ops:
        xs.append(s * a0)
        xs.append(torch.einsum("rkc,rc->rk", v, a) * float(S.INV_SQRT3))
        xv.append(s[:, :, None] * a[:, None, :])
        xv.append(v * a0[:, :, None])
    return torch.cat(xs, dim=1), torch.cat(xv, dim=1)


def o3_tensor_product(p, ops: List[SV], attr: torch.Tensor) -> SV:
    xs, xv = tp_inputs(ops, attr)
    scale = 1.0 / np.sqrt(float(xs.shape[1]))
    s = (xs @ p["ws"]) * scale + p["b"] if p["ws"].shape[1] > 0 else xs.new_zeros((xs.shape[0], 0))
    v = torch.einsum("rkc,km->rmc", xv, p["wv"]) * scale
    return s, v


def gate(x: SV) -> SV:
    s, v = x
    n_act = s.shape[1] - v.shape[1]
    return float(S.C_SILU) * _silu(s[:, :n_act]), v * (float(S.C_SIGMOID) * torch.sigmoid(s[:, n_act:]))[:, :, None]


def segnn_apply_torch(p, node: SV, node_attr, edge_attr, msg: SV, senders, receivers, dim: int, blocks: int, layers: int):
    """segnn_oracle.segnn_apply after the transform; returns the (n, dim) normalised accelerations."""
    n = node[0].shape[0]
    f = o3_tensor_product(p["embedding_nodes"], [node], node_attr)
    for k in range(layers):
        m: List[SV] = [(f[0][senders], f[1][senders]), (f[0][receivers], f[1][receivers]), msg]
        for i in range(blocks):
            m = [gate(o3_tensor_product(p[f"layer_{k}/message_{i}"], m, edge_attr))]
        agg_s = torch.zeros((n, m[0][0].shape[1]), dtype=f[0].dtype).index_add_(0, receivers, m[0][0])
        agg_v = torch.zeros((n,) + tuple(m[0][1].shape[1:]), dtype=f[0].dtype).index_add_(0, receivers, m[0][1])
        x: List[SV] = [f, (agg_s, agg_v)]
        for i in range(blocks - 1):
            x = [gate(o3_tensor_product(p[f"layer_{k}/update_{i}"], x, node_attr))]
        us, uv = o3_tensor_product(p[f"layer_{k}/update_{blocks - 1}"], x, node_attr)
        f = (f[0] + us, f[1] + uv)
    h = [f]
    for i in range(blocks):
        h = [gate(o3_tensor_product(p[f"readout_{i}"], h, node_attr))]
    _, ov = o3_tensor_product(p["output"], h, node_attr)
    return ov[:, 0, :dim]


def inputs_from_features(features, particle_type, n_vels: int, homogeneous: bool, dtype=torch.float64):
    """NumPy transform (no gradient) -> torch tensors."""
    with S.precision(np.float64 if dtype == torch.float64 else np.float32):
        node, node_attr, edge_attr, msg, snd, rcv, dim = S.segnn_transform(features, particle_type, n_vels, homogeneous)
    t = lambda x: torch.tensor(np.asarray(x), dtype=dtype)  # noqa: E731
    return ((t(node.s), t(node.v)), t(node_attr), t(edge_attr), (t(msg.s), t(msg.v)),
            torch.tensor(snd, dtype=torch.long), torch.tensor(rcv, dtype=torch.long), dim)
