"""SEGNN behind the reference's model API - lagrangebench/models/segnn.py:403-610.

Construction arguments are the reference's (segnn.py:444-459); irreps are given as e3nn-style
strings ("5x1o+9x0e").  Two device paths:

* the configuration every published SEGNN config uses - ``lmax_hidden = lmax_attributes = 1``,
  ``scalar_units = 64`` (hidden irreps 32x0e+32x1o through weight_balanced_irreps,
  segnn.py:365-400), ``norm=None`` - runs on the fused kernels of csrc/lb_segnn*.hip.
  Parameters: ``{block: {"ws": (K, Ms), "wv": (K, Mv), "b": (Ms,)}}`` with blocks
  ``embedding_nodes``, ``layer_{k}/message_{i}``, ``layer_{k}/update_{i}``, ``readout_{i}``,
  ``output``; K indexes the tensor-product channels operand by operand, scalar-derived first;
* everything else with ``lmax_hidden, lmax_attributes <= 2`` and ``norm`` in None / "instance" /
  "batch" (``model.generic``) runs on csrc/lb_segnn_gen.hip.  Parameters:
  ``{block: {"w0": (K_0, M_0), "w1": .., "w2": .., "b": (M_0,)}}`` - one matrix per output irrep
  with the rows in e3nn's own order (x chunk major, attribute l minor) - plus, with norm,
  ``layer_{k}/norm_nodes`` (and ``layer_{k}/norm_msg`` for "batch"): ``{"weight", "bias"}``.

``embed_msg_features=True`` is not built (the reference's own code path for it never applies the
embedding: segnn.py:208-214 builds the module and discards it).  oracle/segnn_oracle.py and
oracle/segnn_irreps_oracle.py document the e3nn conventions assumed; parity with e3nn-jax itself
is unpinned - it cannot be installed here.
"""
from __future__ import annotations

import re
from typing import Dict, List, Tuple

import numpy as np

from .._lib import SegnnDesc
from .base import BaseModel


def parse_irreps(irreps) -> Tuple[int, int]:
    """"5x1o + 1x1o + 9x0e" -> (n_scalars 0e, n_vectors 1o); anything else is not built."""
    ns = nv = 0
    for term in str(irreps).replace(" ", "").split("+"):
        if not term:
            continue
        m = re.fullmatch(r"(?:(\d+)x)?(\d+)([eo])", term)
        if m is None:
            raise ValueError(f"cannot parse irreps term {term!r}")
        mul, l, p = int(m.group(1) or 1), int(m.group(2)), m.group(3)
        if (l, p) == (0, "e"):
            ns += mul
        elif (l, p) == (1, "o"):
            nv += mul
        else:
            raise NotImplementedError(f"irrep {l}{p} is not built (0e and 1o only)")
    return ns, nv


def node_irreps(metadata, input_seq_length: int, has_external_force: bool, has_magnitudes: bool,
                has_homogeneous_particles: bool) -> str:
    """models/utils.py:75-97 - irreps of the node features for a dataset."""
    irreps = [f"{input_seq_length - 1}x1o"]
    if not any(metadata["periodic_boundary_conditions"]):
        irreps.append("2x1o")
    if has_external_force:
        irreps.append("1x1o")
    if has_magnitudes:
        irreps.append(f"{input_seq_length - 1}x0e")
    if not has_homogeneous_particles:
        irreps.append("9x0e")
    return "+".join(irreps)


def parse_chunks(irreps) -> List[Tuple[int, int]]:
    """"5x1o+9x0e+2e" -> [(5, 1), (9, 0), (1, 2)] in the order given; spherical-harmonics parity only."""
    out = []
    for term in str(irreps).replace(" ", "").split("+"):
        if not term:
            continue
        m = re.fullmatch(r"(?:(\d+)x)?(\d+)([eo])", term)
        if m is None:
            raise ValueError(f"cannot parse irreps term {term!r}")
        mul, l, p = int(m.group(1) or 1), int(m.group(2)), m.group(3)
        if p != ("e" if l % 2 == 0 else "o") or l > 2:
            raise NotImplementedError(f"irrep {l}{p} is not built (0e, 1o, 2e)")
        out.append((mul, l))
    return out


def path_ok(l1: int, l2: int, l3: int) -> bool:
    """e3nn.tensor_product keeps l3 in l1 x l2 with matching parity; every irrep here has parity (-1)^l."""
    return abs(l1 - l2) <= l3 <= l1 + l2 and (l1 + l2 + l3) % 2 == 0


def weight_balanced_hidden(scalar_units: int, lmax_hidden: int = 1, lmax_attributes: int = 1) -> int:
    """weight_balanced_irreps (segnn.py:365-400): hidden irreps n x (0e + 1o + .. + lmax_hidden) with the smallest n whose
    tensor product with the attributes has >= scalar_units^2 weights (lmax 1 / 1: 4 n^2 >= units^2)."""
    paths = sum(1 for l1 in range(lmax_hidden + 1) for l2 in range(lmax_attributes + 1) for l3 in range(lmax_hidden + 1)
                if path_ok(l1, l2, l3))
    n = 1
    while paths * n * n < scalar_units**2:
        n += 1
    return n


_MSG_CHUNKS = [(1, 1), (1, 0)]   # additional message features "1x1o+1x0e"
_NORMS = {None: 0, "none": 0, "instance": 1, "batch": 2}


class SEGNN(BaseModel):
    def __init__(self, node_features_irreps, edge_features_irreps, scalar_units: int, lmax_hidden: int,
                 lmax_attributes: int, output_irreps, num_mp_steps: int, n_vels: int,
                 velocity_aggregate: str = "avg", homogeneous_particles: bool = True, norm=None,
                 blocks_per_step: int = 2, embed_msg_features: bool = False):
        assert velocity_aggregate in ["avg", "last"], \
            "Invalid velocity aggregate. Must be one of 'avg', 'sum' or 'last'."
        if not (0 <= lmax_hidden <= 2 and 0 <= lmax_attributes <= 2):
            raise NotImplementedError("SEGNN: lmax_hidden / lmax_attributes up to 2 are built")
        assert norm in ["batch", "instance", "none", None], f"Unknown norm '{norm}'"
        if embed_msg_features:
            raise NotImplementedError("SEGNN: embed_msg_features is not built")
        if parse_irreps(output_irreps) != (0, 1):
            raise NotImplementedError("SEGNN: only output_irreps='1x1o' is built")
        if parse_irreps(edge_features_irreps) != (1, 1):
            raise NotImplementedError("SEGNN: edge_features_irreps must be '1x1o+1x0e'")
        self._node_ns, self._node_nv = parse_irreps(node_features_irreps)
        self._node_irreps_str = str(node_features_irreps)
        self._node_chunks = parse_chunks(node_features_irreps)
        self._lmax_hidden, self._lmax_attributes, self._norm = lmax_hidden, lmax_attributes, _NORMS[norm]
        self.norm_eps = 1e-5
        self._hidden = weight_balanced_hidden(scalar_units, lmax_hidden, lmax_attributes)
        # the fused kernels cover hidden 32x0e+32x1o without norm; anything else takes csrc/lb_segnn_gen.hip
        self.generic = not (self._hidden == 32 and lmax_hidden == 1 and lmax_attributes == 1 and self._norm == 0)
        self._num_mp_steps = num_mp_steps
        self._blocks_per_step = blocks_per_step
        self._n_vels = n_vels
        self._velocity_aggregate = velocity_aggregate
        self._homogeneous_particles = homogeneous_particles
        self._handles: Dict[Tuple[int, int], Tuple[object, object]] = {}

    # ------------------------------------------------------------------ parameters
    def block_shapes(self) -> List[Tuple[str, int, int, int]]:
        """(name, K, Ms, Mv) of every O3TensorProduct in call order (= lb_segnn_create's order)."""
        C, B = self._hidden, self._blocks_per_step
        out = [("embedding_nodes", self._node_ns + self._node_nv, C, C)]
        for k in range(self._num_mp_steps):
            for i in range(B):
                out.append((f"layer_{k}/message_{i}", 4 * C + 2 if i == 0 else 2 * C, 2 * C, C))
            for i in range(B):
                out.append((f"layer_{k}/update_{i}", 4 * C if i == 0 else 2 * C, C if i == B - 1 else 2 * C, C))
        for i in range(B):
            out.append((f"readout_{i}", 2 * C, 2 * C, C))
        out.append(("output", 2 * C, 0, 1))
        return out

    # ---- general irreps (model.generic)
    def hidden_chunks(self) -> List[Tuple[int, int]]:
        return [(self._hidden, l) for l in range(self._lmax_hidden + 1)]

    def gen_blocks(self):
        """(name, x chunks, [(mul, l)] outputs ascending in l) of every O3TensorProduct in call order."""
        hid = self.hidden_chunks()
        n, Lh, B = self._hidden, self._lmax_hidden, self._blocks_per_step
        gated = [(n + n * Lh, 0)] + [(n, l) for l in range(1, Lh + 1)]   # one gate scalar per non-scalar irrep
        out = [("embedding_nodes", self._node_chunks, hid)]
        for k in range(self._num_mp_steps):
            for i in range(B):
                out.append((f"layer_{k}/message_{i}", hid + hid + _MSG_CHUNKS if i == 0 else hid, gated))
            for i in range(B):
                out.append((f"layer_{k}/update_{i}", hid + hid if i == 0 else hid, hid if i == B - 1 else gated))
        for i in range(B):
            out.append((f"readout_{i}", hid, gated))
        out.append(("output", hid, [(1, 1)]))
        return out

    def gen_leaves(self) -> List[Tuple[str, str, Tuple[int, ...]]]:
        """(block, leaf, shape) in lb_segnn_create's blob order (layout 2 of include/lbhip.h)."""
        La = self._lmax_attributes
        leaves = []
        for name, xin, outs in self.gen_blocks():
            for mul, l in outs:
                K = sum(m for m, l1 in xin for l2 in range(La + 1) if path_ok(l1, l2, l))
                if K and mul:
                    leaves.append((name, f"w{l}", (K, mul)))
            if outs[0][1] == 0:
                leaves.append((name, "b", (outs[0][0],)))
        if self._norm:
            nw, nb = self._hidden * (self._lmax_hidden + 1), self._hidden
            for k in range(self._num_mp_steps):
                for which in (["norm_msg"] if self._norm == 2 else []) + ["norm_nodes"]:
                    leaves += [(f"layer_{k}/{which}", "weight", (nw,)), (f"layer_{k}/{which}", "bias", (nb,))]
        return leaves

    def init_params(self, seed) -> Dict:
        """uniform_init with weight_std 1 (segnn.py:30-41 under e3nn's "element" normalisation),
        zero biases; BatchNorm weight 1 / bias 0."""
        rng = np.random.default_rng(seed)
        p = {}
        if self.generic:
            for blk, leaf, shape in self.gen_leaves():
                if leaf.startswith("w") and leaf != "weight":
                    v = rng.uniform(-1, 1, size=shape)
                else:
                    v = np.ones(shape) if leaf == "weight" else np.zeros(shape)
                p.setdefault(blk, {})[leaf] = v.astype(np.float32)
            return p
        for name, K, ms, mv in self.block_shapes():
            p[name] = {"ws": rng.uniform(-1, 1, size=(K, ms)).astype(np.float32),
                       "wv": rng.uniform(-1, 1, size=(K, mv)).astype(np.float32),
                       "b": np.zeros((ms,), np.float32)}
        return p

    def init(self, key, sample):
        seed = int(np.asarray(key).ravel()[-1]) if key is not None else 0
        return self.init_params(seed), {}

    def flatten(self, params) -> np.ndarray:
        out = []
        if self.generic:
            for blk, leaf, shape in self.gen_leaves():
                v = np.asarray(params[blk][leaf], np.float32)
                if v.shape != shape:
                    raise ValueError(f"SEGNN params[{blk!r}][{leaf!r}]: expected {shape}, got {v.shape}")
                out.append(v.ravel())
            return np.concatenate(out)
        for name, K, ms, mv in self.block_shapes():
            blk = params[name]
            ws, wv, b = (np.asarray(blk[k], np.float32) for k in ("ws", "wv", "b"))
            if ws.shape != (K, ms) or wv.shape != (K, mv) or b.shape != (ms,):
                raise ValueError(f"SEGNN params[{name!r}]: expected ws {(K, ms)}, wv {(K, mv)}, b {(ms,)}; "
                                 f"got {ws.shape}, {wv.shape}, {b.shape}")
            out += [ws.ravel(), wv.ravel(), b.ravel()]
        return np.concatenate(out)

    def unflatten(self, blob, like=None) -> Dict:
        """Inverse of flatten: a flat blob (weights, gradients or AdamW moments of the training handle) -> parameter dict."""
        blob = np.asarray(blob, np.float32)
        out, o = {}, 0
        if self.generic:
            for blk, leaf, shape in self.gen_leaves():
                n = int(np.prod(shape))
                out.setdefault(blk, {})[leaf] = blob[o:o + n].reshape(shape).copy()
                o += n
            if o != blob.size:
                raise ValueError(f"SEGNN.unflatten: blob has {blob.size} floats, the model {o}")
            return out
        for name, K, ms, mv in self.block_shapes():
            blk = {}
            for leaf, shape in (("ws", (K, ms)), ("wv", (K, mv)), ("b", (ms,))):
                n = int(np.prod(shape))
                blk[leaf] = blob[o:o + n].reshape(shape).copy()
                o += n
            out[name] = blk
        if o != blob.size:
            raise ValueError(f"SEGNN.unflatten: blob has {blob.size} floats, the model {o}")
        return out

    def _desc(self) -> SegnnDesc:
        d = SegnnDesc()
        d.hidden, d.blocks_per_step, d.num_mp_steps = self._hidden, self._blocks_per_step, self._num_mp_steps
        d.homogeneous = int(bool(self._homogeneous_particles))
        d.n_vels = self._n_vels
        d.velocity_avg = int(self._velocity_aggregate == "avg")
        d.lmax_hidden, d.lmax_attributes, d.norm, d.norm_eps = (self._lmax_hidden, self._lmax_attributes, self._norm,
                                                                float(self.norm_eps))
        return d

    def train_handle(self, engine, params):
        """Device-resident training state for `params` on `engine` (csrc/lb_train_segnn.h, round 5)."""
        if self.generic:
            raise NotImplementedError("SEGNN training is built for the shipped configuration (scalar_units 64, lmax 1, norm None)")
        return engine.segnn_train_create(self._desc(), self.flatten(params))

    # ------------------------------------------------------------------ engine binding
    def handle(self, engine, params):
        key = (id(engine), id(params))
        hit = self._handles.get(key)
        if hit is not None and hit[1] is params and hit[0].engine is engine:
            self._handles[key] = self._handles.pop(key)
            return hit[0]
        self._handles.pop(key, None)
        while len(self._handles) >= 4:  # LRU: at most four device copies per model object (models/gns.py)
            self._handles.pop(next(iter(self._handles)))
        h = engine.segnn_create(self._desc(), self.flatten(params))
        self._handles[key] = (h, params)
        return h

    def apply(self, params, state, sample):
        features, particle_type = sample
        engine = getattr(features, "engine", None)
        if engine is None:
            raise TypeError("SEGNN.apply needs the FeatureDict returned by case.preprocess_eval/"
                            "allocate_eval (it names the engine state to run on)")
        if features.version != engine.version:
            raise RuntimeError("features are stale: the engine state changed since they were produced")
        acc = engine.segnn_forward(self.handle(engine, params))
        return {"acc": acc if features.batched else acc[0]}, state

    def __call__(self, params, state, sample):
        return self.apply(params, state, sample)
