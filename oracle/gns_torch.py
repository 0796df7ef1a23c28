"""Differentiable GNS forward in torch (models/gns.py:65-171 through haiku / jraph): TEST INFRASTRUCTURE - the
checker of the device training step (csrc/lb_train.hip: lb_gns_train_loss_grad).  torch.autograd over this
restatement, on the graph (receiver-sorted edge list) and the features the HIP engine built, gives the reference
gradients of trainer.py:35-60's _mse; `tests/test_train.py` checks this file against the NumPy oracle and its
gradients against finite differences, then the engine's gradients against it.  Only tests import it (round 2 had
the Trainer differentiate it; round 3's Trainer steps through the C ABI).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from lagrangebench_amd.models.gns import layer_names


def params_to_torch(params, device=None, requires_grad: bool = False) -> Dict[str, Dict[str, torch.Tensor]]:
    out = {}
    for mod, leaves in params.items():
        out[mod] = {k: torch.tensor(np.asarray(v, np.float32), device=device, requires_grad=requires_grad)
                    for k, v in leaves.items()}
    return out


def params_to_numpy(params_t) -> Dict[str, Dict[str, np.ndarray]]:
    return {mod: {k: v.detach().cpu().numpy().astype(np.float32) for k, v in leaves.items()}
            for mod, leaves in params_t.items()}


def _mlp(p, name: str, x: torch.Tensor, blocks: int) -> torch.Tensor:
    for li in range(blocks):
        lin = p[f"{name}/linear_{li}"]
        x = torch.addmm(lin["b"], x, lin["w"])
        if li < blocks - 1:
            x = torch.relu(x)
    ln = p.get(f"{name}/layer_norm")
    if ln is not None:
        x = torch.nn.functional.layer_norm(x, (x.shape[-1],), ln["scale"], ln["offset"], 1e-5)
    return x


def gns_apply_torch(params_t, node_feats: torch.Tensor, edge_feats: torch.Tensor, senders: torch.Tensor,
                    receivers: torch.Tensor, particle_type: torch.Tensor, num_mp_steps: int,
                    blocks_per_step: int = 2) -> torch.Tensor:
    """nodes (n, F) fp32, edges (E, dim+1) fp32 over REAL edges only (padding removed by the caller),
    senders / receivers (E,) int64 -> normalised accelerations (n, dim)."""
    n = node_feats.shape[0]
    if "embed" in params_t:
        pt = torch.where(particle_type < 0, particle_type + 9, particle_type).long()
        node_feats = torch.cat([node_feats, params_t["embed"]["embeddings"][pt]], dim=-1)
    nl = _mlp(params_t, "enc_node", node_feats, blocks_per_step)
    el = _mlp(params_t, "enc_edge", edge_feats, blocks_per_step)
    names = layer_names(num_mp_steps)
    assert names[0] == "enc_node"
    for k in range(num_mp_steps):
        e2 = _mlp(params_t, f"proc{k}_edge", torch.cat([nl[senders], nl[receivers], el], dim=-1), blocks_per_step)
        agg = torch.zeros((n, e2.shape[1]), dtype=e2.dtype, device=e2.device).index_add_(0, receivers, e2)
        n2 = _mlp(params_t, f"proc{k}_node", torch.cat([nl, agg], dim=-1), blocks_per_step)
        nl = n2 + nl
        el = e2 + el
    return _mlp(params_t, "decoder", nl, blocks_per_step)


def gns_inputs_from_features(features, particle_type, b: int = None):
    """FeatureDict (engine-backed or plain dict) of one trajectory -> the tensors gns_apply_torch takes.
    Column order of GNS._transform (gns.py:135-157); padded edges (index >= n) are dropped."""
    def pick(k):
        v = features[k]
        v = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v))
        return v if b is None else v[b]
    n = pick("vel_hist").shape[0]
    node = torch.cat([pick(k).reshape(n, -1) for k in ("vel_hist", "vel_mag", "bound", "force") if k in features], dim=-1)
    snd, rcv = pick("senders").long(), pick("receivers").long()
    real = rcv < n
    edge = torch.cat([pick("rel_disp"), pick("rel_dist")], dim=-1)[real]
    pt = particle_type if isinstance(particle_type, torch.Tensor) else torch.as_tensor(np.asarray(particle_type))
    pt = pt if b is None else pt[b]
    return node.float(), edge.float(), snd[real], rcv[real], pt.to(node.device)
