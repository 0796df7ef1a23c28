"""CPU oracle for SEGNN (lagrangebench/models/segnn.py:30-610) - TEST INFRASTRUCTURE ONLY.

**Parity unpinned.**  SEGNN's arithmetic lives in e3nn-jax 0.20.3 (poetry.lock:620-621), which is
not installable here, and the reference pins it only through an equivariance property
(tests/models_test.py:70-87), not through numbers.  This file restates the network for the
configuration the reference ships (lmax_hidden = lmax_attributes = 1, configs/*/segnn.yaml) using
the following published e3nn-jax conventions, each an ASSUMPTION that could not be executed here:

 A1  irreps 1o are stored in (x, y, z) order; an IrrepsArray "mul x 1o" chunk is (mul, 3) row-major.
 A2  spherical_harmonics(l<=1, normalize=True, normalization="integral"):
       Y0 = 1/(2 sqrt(pi)),  Y1 = sqrt(3/(4 pi)) * r/|r|  (zero vector -> 0).
 A3  tensor_product(x, y) with irrep_normalization="component": the paths reachable for l <= 1 are
       0e x 0e -> 0e : s a0          1o x 0e -> 1o : v a0
       0e x 1o -> 1o : s a           1o x 1o -> 0e : (v . a) / sqrt(3)
     (1o x 1o -> 1e, 2e are produced by e3nn but dropped by the Linear: no matching output irrep);
     output chunks are emitted per (x chunk, y chunk, ir_out) and regrouped with a stable sort by irrep.
 A4  e3nn.haiku.Linear(path_normalization="element", gradient_normalization="element"): one weight
     (mul_in, mul_out) per irrep type, forward  y = x @ w / sqrt(mul_in)  (+ bias on 0e), weights
     initialised U(-1, 1) through the reference's uniform_init (segnn.py:30-41).
 A5  e3nn.gate(x, even_act=silu, even_gate_act=sigmoid (default), normalize_act=True): the first
     scalars are activated, the LAST `n_vectors` scalars gate the vectors; each activation f is
     rescaled by c_f = 1/sqrt(E_{z~N(0,1)}[f(z)^2]) (c_silu ~ 1.6766, c_sigmoid ~ 1.8463).
 A6  jraph.GraphNetwork passes (edges, sender nodes, receiver nodes, globals) to the edge function:
     the message input is [f_sender | f_receiver | (rel_disp, rel_dist)] (segnn.py:287-291).

What IS checked (tests/test_segnn.py): O(3)-equivariance of this restatement (the reference's own
test for SEGNN), and HIP-vs-this-oracle agreement to 1e-5.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

# Working precision of the restatement: float32 (what the reference computes in).  tests/ switch it to float64
# (`with precision(np.float64):`) to obtain a yardstick for ELEMENT-WISE errors - the same network, constants and
# operation order in double precision; the named constants below stay the float32 values the engine also holds.
F = np.float32


class precision:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global F
        self.prev, F = F, self.dtype
        return self

    def __exit__(self, *exc):
        global F
        F = self.prev
        return False


Y0 = np.float32(1.0 / (2.0 * np.sqrt(np.pi)))
Y1 = np.float32(np.sqrt(3.0 / (4.0 * np.pi)))
INV_SQRT3 = np.float32(1.0 / np.sqrt(3.0))


def _second_moment_const(f) -> np.float32:
    """c = 1/sqrt(E[f(z)^2]), z ~ N(0,1), on e3nn's deterministic quantile grid (A5)."""
    from scipy.special import erfinv
    n = 1_000_001
    z = np.sqrt(2.0) * erfinv(np.linspace(-1.0, 1.0, n + 2)[1:-1])
    return np.float32(1.0 / np.sqrt(np.mean(f(z) ** 2)))


def _silu(x):
    return x / (1.0 + np.exp(-x))


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


C_SILU = _second_moment_const(_silu)
C_SIGMOID = _second_moment_const(_sigmoid)


class SV:
    """A batch of irreps features with l <= 1: scalars (R, ns) and vectors (R, nv, 3)."""

    def __init__(self, s: np.ndarray, v: np.ndarray):
        self.s, self.v = s.astype(F), v.astype(F)

    @property
    def ns(self):
        return self.s.shape[1]

    @property
    def nv(self):
        return self.v.shape[1]

    def __getitem__(self, idx):
        return SV(self.s[idx], self.v[idx])


def cat(parts: List[SV]) -> List[SV]:
    """e3nn.concatenate keeps the operands' chunks in order; TP/Linear below consume the list."""
    return list(parts)


def spherical_harmonics(vec: np.ndarray) -> np.ndarray:
    """(R, 3) -> (R, 4) = [Y0, Y1 * unit vector]  (A2)."""
    vec = vec.astype(F)
    nrm = np.sqrt(np.sum(vec * vec, axis=-1, keepdims=True, dtype=F))
    unit = vec / np.where(nrm == 0, F(1), nrm)
    return np.concatenate([np.full((len(vec), 1), Y0, F), Y1 * unit], axis=-1).astype(F)


def tp_inputs(ops: List[SV], attr: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Tensor product of [operands] with the attribute (a0, a) (A3), regrouped:
    scalar channels (R, K) and vector channels (R, K, 3), K = sum(ns + nv).
    Channel order = operand order, within an operand: scalar-derived first, then vector-derived
    (for an e3nn checkpoint the rows of the weights are permuted to this order at import time)."""
    a0, a = attr[:, :1].astype(F), attr[:, 1:4].astype(F)
    xs, xv = [], []
    for op in ops:
        xs.append(op.s * a0)                                          # 0e x 0e -> 0e
        xs.append(np.einsum("rkc,rc->rk", op.v, a) * INV_SQRT3)       # 1o x 1o -> 0e
        xv.append(op.s[:, :, None] * a[:, None, :])                   # 0e x 1o -> 1o
        xv.append(op.v * a0[:, :, None])                              # 1o x 0e -> 1o
    return np.concatenate(xs, axis=1).astype(F), np.concatenate(xv, axis=1).astype(F)


def o3_tensor_product(p: Dict[str, np.ndarray], ops: List[SV], attr: np.ndarray) -> SV:
    """O3TensorProduct (segnn.py:44-128): tensor_product + Linear (A4).  p: ws (K, Ms), wv (K, Mv),
    b (Ms,)."""
    xs, xv = tp_inputs(ops, attr)
    K = F(xs.shape[1])
    scale = F(1.0) / np.sqrt(K)
    s = (xs @ p["ws"]) * scale + p["b"] if p["ws"].shape[1] > 0 else np.zeros((len(xs), 0), F)
    # (R, K, 3) x (K, Mv) -> (R, Mv, 3) as one BLAS product per row block (the c_einsum form of the same contraction
    # took 27 of the 33 s of a DAM2D-size forward)
    v = np.matmul(np.ascontiguousarray(xv.transpose(0, 2, 1)), p["wv"]).transpose(0, 2, 1) * scale
    return SV(s, v)


def gate(x: SV) -> SV:
    """e3nn.gate (A5): x has ns = n_out_scalars + nv scalars; the last nv scalars gate the vectors."""
    n_act = x.ns - x.nv
    s = C_SILU * _silu(x.s[:, :n_act].astype(F))
    g = C_SIGMOID * _sigmoid(x.s[:, n_act:].astype(F))
    return SV(s, x.v * g[:, :, None])


def init_tp(rng: np.random.Generator, K: int, ms: int, mv: int) -> Dict[str, np.ndarray]:
    """uniform_init with weight_std = 1 (A4); biases start at 0 (e3nn Linear default)."""
    return {"ws": rng.uniform(-1, 1, size=(K, ms)).astype(F),
            "wv": rng.uniform(-1, 1, size=(K, mv)).astype(F),
            "b": np.zeros((ms,), F)}


def weight_balanced_hidden(scalar_units: int) -> int:
    """weight_balanced_irreps for lmax 1 (segnn.py:365-400): 4 n^2 >= scalar_units^2."""
    n = 0
    while 4 * n * n < scalar_units**2:
        n += 1
    return n


def segnn_init(rng: np.random.Generator, node_ns: int, node_nv: int, num_mp_steps: int = 10,
               scalar_units: int = 64, blocks_per_step: int = 2, random_bias: bool = False):
    """Parameters of SEGNN(lmax 1): embedding, per layer message tp_0..tp_{B-1} and update
    tp_0..tp_{B-1}, decoder readout_0..readout_{B-1} and output (segnn.py:184-249,252-362,595-610)."""
    C = weight_balanced_hidden(scalar_units)
    p = {"hidden": C, "blocks": blocks_per_step, "layers": num_mp_steps}
    p["embedding_nodes"] = init_tp(rng, node_ns + node_nv, C, C)
    for k in range(num_mp_steps):
        kin = 2 * (C + C) + (1 + 1)  # [f_s | f_r | (1x1o + 1x0e)]
        for i in range(blocks_per_step):
            p[f"layer_{k}/message_{i}"] = init_tp(rng, kin if i == 0 else 2 * C, 2 * C, C)  # gated: C gates
        for i in range(blocks_per_step):
            kin_u = 2 * (C + C) if i == 0 else 2 * C
            last = i == blocks_per_step - 1
            p[f"layer_{k}/update_{i}"] = init_tp(rng, kin_u, C if last else 2 * C, C)
    for i in range(blocks_per_step):
        p[f"readout_{i}"] = init_tp(rng, 2 * C, 2 * C, C)
    p["output"] = init_tp(rng, 2 * C, 0, 1)
    if random_bias:
        for k, v in p.items():
            if isinstance(v, dict) and v["b"].size:
                v["b"] = rng.uniform(-0.5, 0.5, size=v["b"].shape).astype(F)
    return p


def segnn_transform(features: Dict[str, np.ndarray], particle_type: np.ndarray, n_vels: int,
                    homogeneous: bool, velocity_aggregate: str = "avg"):
    """SEGNN._transform (segnn.py:513-587): node features SV, node / edge attributes, message feats."""
    n = features["vel_hist"].shape[0]
    dim = features["vel_hist"].shape[1] // n_vels

    def pad3(x):  # features_2d_to_3d (models/utils.py:118-138)
        x = np.asarray(x, F)
        if dim == 3:
            return x
        return np.concatenate([x, np.zeros(x.shape[:-1] + (1,), F)], axis=-1)

    vel_hist = pad3(np.asarray(features["vel_hist"], F).reshape(n, n_vels, dim))
    rel_disp = pad3(features["rel_disp"])
    vel = vel_hist.mean(axis=1) if velocity_aggregate == "avg" else vel_hist[:, -1]
    if n_vels == 1:
        vel = vel_hist[:, 0]
    senders = np.asarray(features["senders"]).astype(np.int64)
    receivers = np.asarray(features["receivers"]).astype(np.int64)
    real = receivers < n
    senders, receivers, rel_disp = senders[real], receivers[real], rel_disp[real]
    rel_dist = np.asarray(features["rel_dist"], F)[real]
    edge_attr = spherical_harmonics(rel_disp)
    vel_emb = spherical_harmonics(vel)
    cnt = np.maximum(np.bincount(receivers, minlength=n), 1).astype(F)
    scat = np.zeros((n, 4), F)
    np.add.at(scat, receivers, edge_attr)
    node_attr = vel_emb + scat / cnt[:, None]
    node_attr[:, 0] = 1.0
    vecs = [vel_hist]
    if "bound" in features:
        b = np.asarray(features["bound"], F)
        vecs.append(pad3(np.stack([b[:, :dim], b[:, dim:]], axis=1)))
    if "force" in features:
        vecs.append(pad3(np.asarray(features["force"], F))[:, None, :])
    scal = []
    if "vel_mag" in features:
        scal.append(np.asarray(features["vel_mag"], F))
    if not homogeneous:
        pt = np.where(particle_type < 0, particle_type + 9, particle_type)
        scal.append(np.eye(9, dtype=F)[pt])
    node = SV(np.concatenate(scal, axis=1) if scal else np.zeros((n, 0), F),
              np.concatenate(vecs, axis=1))
    msg = SV(rel_dist.reshape(-1, 1), rel_disp[:, None, :])
    return node, node_attr.astype(F), edge_attr, msg, senders, receivers, dim


def segnn_apply(p, features, particle_type, n_vels: int, homogeneous: bool, return_latents: bool = False):
    """SEGNN.__call__ (segnn.py:595-610)."""
    node, node_attr, edge_attr, msg, senders, receivers, dim = segnn_transform(
        features, particle_type, n_vels, homogeneous)
    n = node.s.shape[0]
    B = p["blocks"] if "blocks" in p else sum(1 for k in p if k.startswith("readout_"))
    L = p["layers"] if "layers" in p else sum(1 for k in p if k.endswith("/message_0"))
    f = o3_tensor_product(p["embedding_nodes"], [node], node_attr)
    lat = [f]
    for k in range(L):
        m: List[SV] = cat([f[senders], f[receivers], msg])
        for i in range(B):
            m = [gate(o3_tensor_product(p[f"layer_{k}/message_{i}"], m, edge_attr))]
        agg_s = np.zeros((n, m[0].ns), F)
        agg_v = np.zeros((n, m[0].nv, 3), F)
        np.add.at(agg_s, receivers, m[0].s)
        np.add.at(agg_v, receivers, m[0].v)
        x: List[SV] = cat([f, SV(agg_s, agg_v)])
        for i in range(B - 1):
            x = [gate(o3_tensor_product(p[f"layer_{k}/update_{i}"], x, node_attr))]
        upd = o3_tensor_product(p[f"layer_{k}/update_{B - 1}"], x, node_attr)
        f = SV(f.s + upd.s, f.v + upd.v)
        lat.append(f)
    h = [f]
    for i in range(B):
        h = [gate(o3_tensor_product(p[f"readout_{i}"], h, node_attr))]
    out = o3_tensor_product(p["output"], h, node_attr)
    acc = out.v[:, 0, :]
    if dim == 2:
        acc = acc[:, :2]
    if return_latents:
        return {"acc": acc.astype(F)}, lat
    return {"acc": acc.astype(F)}
