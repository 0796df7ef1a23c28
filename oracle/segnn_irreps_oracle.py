"""CPU oracle for SEGNN with general irreps - lmax_hidden / lmax_attributes up to 2 and
norm = None | "instance" | "batch" (lagrangebench/models/segnn.py:44-181,252-400,513-610) -
TEST INFRASTRUCTURE ONLY (tests/ import it; nothing under lagrangebench_amd/ does).

**Parity unpinned.**  Like oracle/segnn_oracle.py (whose assumptions A1 - A6 this file shares and
generalises) the arithmetic lives in e3nn-jax 0.20.3, which cannot be installed here; the reference
pins SEGNN only through an equivariance property (tests/models_test.py:70-87).  Additional
assumptions of THIS file, none of which could be executed:

 A7  real Wigner 3j symbols as e3nn builds them: the SU(2) Clebsch-Gordan coefficients (Racah's
     formula) carried to the real basis by Q_l = (-i)^l * [standard complex <- real change of basis],
     the result normalised to Frobenius norm 1.  m = -l..l is the storage order; for l = 1 that is
     (x, y, z) with y the polar axis (A1).  The SIGN of each symbol follows from the formula and is the
     thing a real checkpoint with lmax > 1 would be sensitive to.
 A8  tensor_product(irrep_normalization="component"): path (l1, l2 -> l3) = sqrt(2 l3 + 1) * w3j
     contraction; paths need |l1 - l2| <= l3 <= l1 + l2 and matching parity.  Every irrep on this path is
     of spherical-harmonics type (parity (-1)^l: 0e, 1o, 2e), so parity reduces to l1 + l2 + l3 even.
     The output is regrouped: sorted by irrep (stable), equal irreps merged - the rows of the Linear's
     matrix for an output irrep follow (x chunk, attribute chunk) loop order.
 A9  spherical_harmonics(l <= 2, normalize=True, normalization="integral") =
       Y2 = 1/(2 sqrt(pi)) * [sqrt15 x z, sqrt15 x y, sqrt5 (y^2 - (x^2 + z^2)/2), sqrt15 y z,
                              sqrt15/2 (z^2 - x^2)]   on the unit vector (zero vector -> 0);
     consistent with A7: Y2 is a POSITIVE multiple of the 1o x 1o -> 2e product of the vector with
     itself (checked in tests/test_segnn_irreps.py).
 A10 e3nn.haiku.BatchNorm(irreps, eps=1e-4? (held as a parameter, default 1e-5 here - see NORM_EPS),
     affine=True, reduce="mean", normalization="component", instance=...) called WITHOUT is_training,
     i.e. with its default True (segnn.py:303,347-351): batch statistics always, running averages
     never read.  Scalars: minus the mean; every irrep: divided by sqrt(mean over the batch of the
     component-mean square + eps), times weight; scalars: plus bias.  With instance=True the
     statistics run over the axes BETWEEN the first and the last - for the (N, dim) node array
     that is an axis of length one: scalars become exactly `bias`, every vector/tensor channel is
     normalised per node.  (Degenerate, but it is what the call computes.)
     Deviation: the reference's "batch" statistics include jraph's PADDING edges / the padding node;
     here they run over the real edges / nodes of a trajectory.

What IS checked (tests/test_segnn_irreps.py): O(3) equivariance, agreement with
oracle/segnn_oracle.py for lmax 1 / norm None, 3j symmetry properties, and HIP-vs-this-oracle 1e-5.
"""
from __future__ import annotations

import math
from fractions import Fraction
from functools import lru_cache
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import segnn_oracle as S1

Chunks = List[Tuple[int, int]]          # [(mul, l)], parity (-1)^l implied
NORM_EPS = 1e-5


# --------------------------------------------------------------------------------- 3j symbols (A7)
def _f(n: int) -> int:
    return math.factorial(n)


def _su2_cg_coeff(j1, m1, j2, m2, j3, m3) -> float:
    """<j1 m1 j2 m2 | j3 m3> by Racah's formula (integer spins)."""
    if m3 != m1 + m2:
        return 0.0
    vmin = max(-j1 + j2 + m3, -j1 + m1, 0)
    vmax = min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3)
    C = Fraction((2 * j3 + 1) * _f(j3 + j1 - j2) * _f(j3 - j1 + j2) * _f(j1 + j2 - j3) * _f(j3 + m3) * _f(j3 - m3),
                 _f(j1 + j2 + j3 + 1) * _f(j1 - m1) * _f(j1 + m1) * _f(j2 - m2) * _f(j2 + m2))
    S = Fraction(0)
    for v in range(vmin, vmax + 1):
        S += Fraction((-1) ** (v + j2 + m2) * _f(j2 + j3 + m1 - v) * _f(j1 - m1 + v),
                      _f(v) * _f(j3 - j1 + j2 - v) * _f(j3 + m3 - v) * _f(v + j1 - j2 - m3))
    return math.sqrt(float(C)) * float(S)


def _real_to_complex(l: int) -> np.ndarray:
    q = np.zeros((2 * l + 1, 2 * l + 1), np.complex128)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = 1 / math.sqrt(2)
        q[l + m, l - abs(m)] = -1j / math.sqrt(2)
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m / math.sqrt(2)
        q[l + m, l - abs(m)] = 1j * (-1) ** m / math.sqrt(2)
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def w3j(l1: int, l2: int, l3: int) -> np.ndarray:
    """Real 3j symbol (2 l1 + 1, 2 l2 + 1, 2 l3 + 1), Frobenius norm 1."""
    C = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1))
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            m3 = m1 + m2
            if abs(m3) <= l3:
                C[l1 + m1, l2 + m2, l3 + m3] = _su2_cg_coeff(l1, m1, l2, m2, l3, m3)
    Q1, Q2, Q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    Cr = np.einsum("ij,kl,mn,ikn->jlm", Q1, Q2, np.conj(Q3.T), C.astype(np.complex128))
    assert np.abs(Cr.imag).max() < 1e-12
    Cr = Cr.real
    out = Cr / np.linalg.norm(Cr)
    out.setflags(write=False)
    return out


def path_ok(l1: int, l2: int, l3: int) -> bool:
    return abs(l1 - l2) <= l3 <= l1 + l2 and (l1 + l2 + l3) % 2 == 0


def cg(l1: int, l2: int, l3: int) -> np.ndarray:
    """Path coefficients of tensor_product under "component" normalisation (A8)."""
    return math.sqrt(2 * l3 + 1) * w3j(l1, l2, l3)


# --------------------------------------------------------------------------------- irreps helpers
def parse(irreps) -> Chunks:
    """"5x1o+9x0e+2e" -> [(5, 1), (9, 0), (1, 2)]; only spherical-harmonics parity is built."""
    import re
    out = []
    for term in str(irreps).replace(" ", "").split("+"):
        if not term:
            continue
        m = re.fullmatch(r"(?:(\d+)x)?(\d+)([eo])", term)
        if m is None:
            raise ValueError(f"cannot parse irreps term {term!r}")
        mul, l, p = int(m.group(1) or 1), int(m.group(2)), m.group(3)
        if p != ("e" if l % 2 == 0 else "o"):
            raise NotImplementedError(f"irrep {l}{p}: only spherical-harmonics parity (0e, 1o, 2e, ..) is built")
        out.append((mul, l))
    return out


def dim_of(chunks: Chunks) -> int:
    return sum(mul * (2 * l + 1) for mul, l in chunks)


def sh_chunks(lmax: int) -> Chunks:
    return [(1, l) for l in range(lmax + 1)]


def weight_balanced_chunks(scalar_units: int, lmax_attr: int, lmax_hidden: int) -> Chunks:
    """weight_balanced_irreps (segnn.py:365-400): n x (0e + 1o + .. + lmax_hidden) with the smallest n whose
    tensor product with the attributes has >= scalar_units^2 weights."""
    n = 0
    while True:
        n += 1
        paths = sum(1 for l1 in range(lmax_hidden + 1) for l2 in range(lmax_attr + 1) for l3 in range(lmax_hidden + 1)
                    if path_ok(l1, l2, l3))
        if paths * n * n >= scalar_units ** 2:
            return [(n, l) for l in range(lmax_hidden + 1)]


def tp_rows(x_chunks: Chunks, lmax_attr: int, l3: int) -> List[Tuple[int, int, int]]:
    """Rows of the Linear's matrix for output irrep l3, in e3nn order: [(x chunk index, l2, mul)]."""
    return [(i, l2, mul) for i, (mul, l1) in enumerate(x_chunks) for l2 in range(lmax_attr + 1) if path_ok(l1, l2, l3)]


def tp_K(x_chunks: Chunks, lmax_attr: int, l3: int) -> int:
    return sum(mul for _, _, mul in tp_rows(x_chunks, lmax_attr, l3))


# --------------------------------------------------------------------------------- spherical harmonics (A2, A9)
def spherical_harmonics(vec: np.ndarray, lmax: int) -> np.ndarray:
    """(R, 3) -> (R, (lmax + 1)^2), integral normalisation on the unit vector."""
    F = S1.F
    if lmax > 2:
        raise NotImplementedError("spherical harmonics beyond l = 2 are not built")
    vec = vec.astype(F)
    nrm = np.sqrt(np.sum(vec * vec, axis=-1, keepdims=True, dtype=F))
    u = vec / np.where(nrm == 0, F(1), nrm)
    x, y, z = u[:, 0], u[:, 1], u[:, 2]
    c = F(1.0 / (2.0 * math.sqrt(math.pi)))
    cols = [np.full(len(vec), c, F)]
    if lmax >= 1:
        s3 = F(math.sqrt(3.0))
        cols += [c * s3 * x, c * s3 * y, c * s3 * z]
    if lmax >= 2:
        s15, s5 = F(math.sqrt(15.0)), F(math.sqrt(5.0))
        cols += [c * s15 * x * z, c * s15 * x * y, c * s5 * (y * y - F(0.5) * (x * x + z * z)), c * s15 * y * z,
                 c * (s15 * F(0.5)) * (z * z - x * x)]
    return np.stack(cols, axis=-1).astype(F)


# --------------------------------------------------------------------------------- tensor product + Linear
def split(x: np.ndarray, chunks: Chunks) -> List[np.ndarray]:
    out, o = [], 0
    for mul, l in chunks:
        d = mul * (2 * l + 1)
        out.append(x[:, o:o + d].reshape(len(x), mul, 2 * l + 1))
        o += d
    return out


def tp_features(x: np.ndarray, x_chunks: Chunks, attr: np.ndarray, lmax_attr: int, l3: int) -> np.ndarray:
    """(R, K_l3, 2 l3 + 1): the rows the Linear contracts for output irrep l3 (A8)."""
    F = S1.F
    xs = split(x.astype(F), x_chunks)
    parts = []
    for i, l2, mul in tp_rows(x_chunks, lmax_attr, l3):
        l1 = x_chunks[i][1]
        a = attr[:, l2 * l2:(l2 + 1) * (l2 + 1)].astype(F)
        parts.append(np.einsum("rui,rj,ijk->ruk", xs[i], a, cg(l1, l2, l3).astype(F)).astype(F))
    if not parts:
        return np.zeros((len(x), 0, 2 * l3 + 1), F)
    return np.concatenate(parts, axis=1)


def o3_tensor_product(p: Dict[str, np.ndarray], x: np.ndarray, x_chunks: Chunks, attr: np.ndarray, lmax_attr: int,
                      out_chunks: Chunks) -> np.ndarray:
    """O3TensorProduct (segnn.py:44-128): tensor_product + e3nn Linear ("element": 1 / sqrt(K_l)); out_chunks sorted
    by l, one chunk per l.  p: "w{l}" (K_l, mul_l), "b" (mul_0,)."""
    F = S1.F
    outs = []
    for mul, l in out_chunks:
        X = tp_features(x, x_chunks, attr, lmax_attr, l)
        K = X.shape[1]
        if K == 0 or mul == 0:   # unreachable output: e3nn's Linear leaves it zero
            outs.append(np.zeros((len(x), mul * (2 * l + 1)), F))
            continue
        y = np.matmul(np.ascontiguousarray(X.transpose(0, 2, 1)), p[f"w{l}"].astype(F)) * (F(1.0) / np.sqrt(F(K)))
        y = y.transpose(0, 2, 1)            # (R, mul, 2l+1)
        if l == 0 and "b" in p and p["b"].size:
            y = y + p["b"].astype(F)[None, :, None]
        outs.append(y.reshape(len(x), -1).astype(F))
    return np.concatenate(outs, axis=1).astype(F)


def gated_chunks(hidden: Chunks) -> Chunks:
    """Output irreps of the tensor product inside O3TensorProductGate (segnn.py:158-167): hidden + one gate scalar
    per non-scalar irrep, regrouped."""
    n0 = sum(mul for mul, l in hidden if l == 0)
    ng = sum(mul for mul, l in hidden if l > 0)
    return [(n0 + ng, 0)] + [(mul, l) for mul, l in hidden if l > 0]


def gate(y: np.ndarray, hidden: Chunks) -> np.ndarray:
    """e3nn.gate (A5) on an array with gated_chunks(hidden) irreps -> hidden irreps."""
    F = S1.F
    n0 = sum(mul for mul, l in hidden if l == 0)
    ng = sum(mul for mul, l in hidden if l > 0)
    s = S1.C_SILU * S1._silu(y[:, :n0].astype(F))
    g = S1.C_SIGMOID * S1._sigmoid(y[:, n0:n0 + ng].astype(F))
    outs, o, gi = [s.astype(F)], n0 + ng, 0
    for mul, l in hidden:
        if l == 0:
            continue
        d = 2 * l + 1
        v = y[:, o:o + mul * d].reshape(len(y), mul, d)
        outs.append((v * g[:, gi:gi + mul, None]).reshape(len(y), -1).astype(F))
        o += mul * d
        gi += mul
    return np.concatenate(outs, axis=1).astype(F)


def batch_norm(x: np.ndarray, chunks: Chunks, weight: np.ndarray, bias: np.ndarray, instance: bool,
               eps: float = NORM_EPS) -> np.ndarray:
    """e3nn BatchNorm in training mode over the rows of x (A10); weight: one per channel of every chunk, bias: one per
    scalar channel."""
    F = S1.F
    outs, o, iw, ib = [], 0, 0, 0
    for mul, l in chunks:
        d = 2 * l + 1
        f = x[:, o:o + mul * d].reshape(len(x), mul, d).astype(F)
        if l == 0:
            mean = f if instance else f.mean(axis=0, keepdims=True, dtype=F)
            f = f - mean
        nrm = np.mean(f * f, axis=2, dtype=F)                    # (R, mul)   "component"
        if not instance:
            nrm = nrm.mean(axis=0, keepdims=True, dtype=F)        # (1, mul)
        scale = (F(1.0) / np.sqrt(nrm + F(eps))) * weight[iw:iw + mul].astype(F)[None, :]
        f = f * scale[:, :, None]
        if l == 0:
            f = f + bias[ib:ib + mul].astype(F)[None, :, None]
            ib += mul
        outs.append(f.reshape(len(x), -1).astype(F))
        o += mul * d
        iw += mul
    return np.concatenate(outs, axis=1).astype(F)


# --------------------------------------------------------------------------------- the network
def node_chunks(n_vels: int, has_bound: bool, has_force: bool, has_mag: bool, homogeneous: bool) -> Chunks:
    """models/utils.py:75-97 (node_irreps)."""
    c = [(n_vels, 1)]
    if has_bound:
        c.append((2, 1))
    if has_force:
        c.append((1, 1))
    if has_mag:
        c.append((n_vels, 0))
    if not homogeneous:
        c.append((9, 0))
    return c


MSG_CHUNKS: Chunks = [(1, 1), (1, 0)]   # additional message features "1x1o+1x0e" (rel_disp, rel_dist)


def block_list(x_node: Chunks, hidden: Chunks, L: int, B: int):
    """(name, x_chunks, out_chunks, kind) of every O3TensorProduct in call order; kind = "plain" | "gate"."""
    blocks = [("embedding_nodes", x_node, hidden, "plain")]
    gated = gated_chunks(hidden)
    for k in range(L):
        for i in range(B):
            xin = hidden + hidden + MSG_CHUNKS if i == 0 else hidden
            blocks.append((f"layer_{k}/message_{i}", xin, gated, "gate"))
        for i in range(B):
            xin = hidden + hidden if i == 0 else hidden
            last = i == B - 1
            blocks.append((f"layer_{k}/update_{i}", xin, hidden if last else gated, "plain" if last else "gate"))
    for i in range(B):
        blocks.append((f"readout_{i}", hidden, gated, "gate"))
    blocks.append(("output", hidden, [(1, 1)], "plain"))
    return blocks


def segnn_init(rng: np.random.Generator, x_node: Chunks, num_mp_steps: int = 10, scalar_units: int = 64,
               lmax_hidden: int = 1, lmax_attr: int = 1, blocks_per_step: int = 2, norm: Optional[str] = None,
               random_bias: bool = False):
    F = S1.F
    hidden = weight_balanced_chunks(scalar_units, lmax_attr, lmax_hidden)
    p = {"hidden": hidden, "blocks": blocks_per_step, "layers": num_mp_steps, "lmax_attr": lmax_attr,
         "norm": norm if norm not in (None, "none") else None, "x_node": list(x_node)}
    for name, xin, out, _ in block_list(x_node, hidden, num_mp_steps, blocks_per_step):
        blk = {}
        for mul, l in out:
            blk[f"w{l}"] = rng.uniform(-1, 1, size=(tp_K(xin, lmax_attr, l), mul)).astype(F)
        n0 = sum(mul for mul, l in out if l == 0)
        blk["b"] = (rng.uniform(-0.5, 0.5, size=(n0,)) if random_bias else np.zeros((n0,))).astype(F)
        p[name] = blk
    if p["norm"]:
        nw = sum(mul for mul, _ in hidden)
        n0 = sum(mul for mul, l in hidden if l == 0)
        for k in range(num_mp_steps):
            names = [f"layer_{k}/norm_nodes"] + ([f"layer_{k}/norm_msg"] if p["norm"] == "batch" else [])
            for nm in names:
                p[nm] = {"weight": (1.0 + 0.2 * rng.standard_normal(nw)).astype(F) if random_bias else np.ones(nw, F),
                         "bias": (0.2 * rng.standard_normal(n0)).astype(F) if random_bias else np.zeros(n0, F)}
    return p


def node_feature_rows(node: "S1.SV", x_node: Chunks) -> np.ndarray:
    """SV (scalars, vectors in the order vel_hist | bound | force; scalars vel_mag | one-hot) -> e3nn rows of x_node."""
    F = S1.F
    outs, iv, is_ = [], 0, 0
    for mul, l in x_node:
        if l == 1:
            outs.append(node.v[:, iv:iv + mul].reshape(len(node.s), -1))
            iv += mul
        elif l == 0:
            outs.append(node.s[:, is_:is_ + mul])
            is_ += mul
        else:
            raise NotImplementedError
    assert iv == node.nv and is_ == node.ns
    return np.concatenate(outs, axis=1).astype(F)


def segnn_apply(p, features, particle_type, n_vels: int, homogeneous: bool, return_latents: bool = False,
                norm_eps: float = NORM_EPS, velocity_aggregate: str = "avg"):
    """SEGNN.__call__ (segnn.py:595-610) for general irreps; features as oracle/segnn_oracle.segnn_transform takes them."""
    F = S1.F
    La, hidden, B, L, norm = p["lmax_attr"], p["hidden"], p["blocks"], p["layers"], p["norm"]
    node, _, _, msg, senders, receivers, dim = S1.segnn_transform(features, particle_type, n_vels, homogeneous,
                                                                  velocity_aggregate)
    n = node.s.shape[0]
    x_node = p["x_node"]
    x = node_feature_rows(node, x_node)
    # attributes (segnn.py:556-575): recomputed here up to lmax_attr
    rel_disp = msg.v[:, 0, :]
    edge_attr = spherical_harmonics(rel_disp, La)
    vh = node.v[:, :n_vels]
    vel = vh.mean(axis=1, dtype=F) if n_vels > 1 else vh[:, 0]
    if n_vels > 1 and velocity_aggregate == "last":
        vel = vh[:, -1]
    cnt = np.maximum(np.bincount(receivers, minlength=n), 1).astype(F)
    scat = np.zeros((n, edge_attr.shape[1]), F)
    np.add.at(scat, receivers, edge_attr)
    node_attr = (spherical_harmonics(vel, La) + scat / cnt[:, None]).astype(F)
    node_attr[:, 0] = 1.0
    msg_rows = np.concatenate([rel_disp, msg.s], axis=1).astype(F)     # "1x1o+1x0e"
    gated = gated_chunks(hidden)

    def tp(name, xin, xc, attr, outc):
        return o3_tensor_product(p[name], xin, xc, attr, La, outc)

    f = tp("embedding_nodes", x, x_node, node_attr, hidden)
    lat = [f]
    for k in range(L):
        m = np.concatenate([f[senders], f[receivers], msg_rows], axis=1)
        mc = hidden + hidden + MSG_CHUNKS
        for i in range(B):
            m = gate(tp(f"layer_{k}/message_{i}", m, mc, edge_attr, gated), hidden)
            mc = hidden
        if norm == "batch":
            q = p[f"layer_{k}/norm_msg"]
            m = batch_norm(m, hidden, q["weight"], q["bias"], False, norm_eps) if len(m) else m
        agg = np.zeros((n, m.shape[1]), F)
        np.add.at(agg, receivers, m)
        xu = np.concatenate([f, agg], axis=1)
        xc = hidden + hidden
        for i in range(B - 1):
            xu = gate(tp(f"layer_{k}/update_{i}", xu, xc, node_attr, gated), hidden)
            xc = hidden
        upd = tp(f"layer_{k}/update_{B - 1}", xu, xc, node_attr, hidden)
        f = (f + upd).astype(F)
        if norm in ("batch", "instance"):
            q = p[f"layer_{k}/norm_nodes"]
            f = batch_norm(f, hidden, q["weight"], q["bias"], norm == "instance", norm_eps)
        lat.append(f)
    h = f
    for i in range(B):
        h = gate(tp(f"readout_{i}", h, hidden, node_attr, gated), hidden)
    out = tp("output", h, hidden, node_attr, [(1, 1)])
    acc = out[:, :3]
    if dim == 2:
        acc = acc[:, :2]
    if return_latents:
        return {"acc": acc.astype(F)}, lat
    return {"acc": acc.astype(F)}
