"""torch-CPU (all host cores) variant of the oracle's GNS forward, used ONLY by bench.py's
``cpu_baseline`` leg ("reference-shaped CPU restatement (torch-CPU)").

Same algorithmic shape as the reference (lagrangebench/models/gns.py:65-171 through haiku/jraph):
MLPs over all E_cap padded rows, unfused gather / Linear / ReLU / Linear / LayerNorm /
index_add (= jraph.segment_sum), fp32.  It is checked against the NumPy oracle in
tests/test_oracle_golden.py; it is test/benchmark infrastructure, never product code.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import lb_oracle as O


def params_to_torch(params) -> Dict[str, Dict[str, torch.Tensor]]:
    return {k: {kk: torch.from_numpy(np.ascontiguousarray(vv)) for kk, vv in v.items()} for k, v in params.items()}


def _mlp(p, name, x, bps):
    for li in range(bps):
        lin = p[f"{name}/linear_{li}"]
        x = torch.addmm(lin["b"], x, lin["w"])
        if li < bps - 1:
            x = torch.relu(x)
    ln = p.get(f"{name}/layer_norm")
    if ln is not None:
        x = torch.nn.functional.layer_norm(x, (x.shape[-1],), ln["scale"], ln["offset"], 1e-5)
    return x


@torch.no_grad()
def gns_apply(params_t, features, particle_type, num_mp_steps=10, blocks_per_step=2,
              skip_padding=False, return_intermediates=False, dtype=torch.float32):
    """GNS.__call__ in the reference's padded shape (padding id N gathers node N-1, is dropped by
    the scatter-add).  ``skip_padding`` evaluates the real edges only (same node outputs);
    ``return_intermediates`` also returns the node latents after the encoder and every layer
    (same keys as lb_oracle.gns_apply), used by the full-size parity tests.  ``dtype=torch.float64`` evaluates the
    same fp32 inputs and weights in double precision: the yardstick the element-wise error tests measure both the
    engine and an fp32 evaluation against."""
    nodes, edges = O.gns_transform(features)
    nodes, edges = torch.from_numpy(nodes).to(dtype), torch.from_numpy(edges).to(dtype)
    if dtype != torch.float32:
        params_t = {k: {kk: vv.to(dtype) for kk, vv in v.items()} for k, v in params_t.items()}
    n = nodes.shape[0]
    senders = torch.from_numpy(np.asarray(features["senders"]).astype(np.int64))
    receivers = torch.from_numpy(np.asarray(features["receivers"]).astype(np.int64))
    if skip_padding:
        keep = receivers < n
        senders, receivers, edges = senders[keep], receivers[keep], edges[keep]
    if "embed" in params_t:
        pt = torch.from_numpy(np.where(particle_type < 0, particle_type + O.NodeType.SIZE, particle_type).astype(np.int64))
        nodes = torch.cat([nodes, params_t["embed"]["embeddings"][pt]], dim=-1)
    nl = _mlp(params_t, "enc_node", nodes, blocks_per_step)
    el = _mlp(params_t, "enc_edge", edges, blocks_per_step)
    sc, rc = senders.clamp(max=n - 1), receivers.clamp(max=n - 1)
    real = receivers < n
    inter = {"enc_n": nl.numpy().copy()} if return_intermediates else None
    for k in range(num_mp_steps):
        ein = torch.cat([nl[sc], nl[rc], el], dim=-1)
        e2 = _mlp(params_t, f"proc{k}_edge", ein, blocks_per_step)
        agg = torch.zeros((n + 1, e2.shape[1]), dtype=e2.dtype)
        agg.index_add_(0, torch.where(real, receivers, torch.full_like(receivers, n)), e2)
        n2 = _mlp(params_t, f"proc{k}_node", torch.cat([nl, agg[:n]], dim=-1), blocks_per_step)
        nl = n2 + nl
        el = e2 + el
        if return_intermediates:
            inter[f"n{k}"] = nl.numpy().copy()
    acc = _mlp(params_t, "decoder", nl, blocks_per_step)
    if return_intermediates:
        return {"acc": acc.numpy()}, inter
    return {"acc": acc.numpy()}
