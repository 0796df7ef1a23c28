"""CPU oracle: a NumPy restatement of the LagrangeBench inference-rollout hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it; nothing under ``lagrangebench_amd/`` does.  The product path is the HIP library
(``lagrangebench_amd/csrc``) and fails loudly when that library is missing.

What is restated (file:line relative to /root/reference):

=====================  ==========================================================
``space_periodic/free``  jax_sph.jax_md.space (3rd party, jax-sph 0.0.3, pinned in
                         poetry.lock:1137) as used by case_setup/case.py:104-108
``get_dataset_stats``    lagrangebench/data/utils.py:9-45
``neighbor_list``        jax_sph.jax_md.partition.neighbor_list (3rd party; call
                         site case_setup/case.py:120-130, .allocate :184-186,
                         .update :188-190) - backend "jaxmd_vmap", Sparse format
``feature_transform``    lagrangebench/case_setup/features.py:47-126
``case_builder``         lagrangebench/case_setup/case.py:62-269
``GNS``                  lagrangebench/models/gns.py:35-171, models/utils.py:100-115
                         (haiku nets.MLP / LayerNorm / Embed, jraph GraphNetwork,
                         jraph.segment_sum: 3rd party, dm-haiku 0.0.12 / jraph 0.0.6.dev0)
``forward_eval``         lagrangebench/evaluate/rollout.py:31-75
``eval_batched_rollout`` lagrangebench/evaluate/rollout.py:78-178
``mse / mae``            lagrangebench/evaluate/metrics.py:86-96,139-147
``get_kinematic_mask``   lagrangebench/utils.py:28-35
=====================  ==========================================================

Pinning status
--------------
JAX/Haiku/jraph/jax-sph are not installable in the build container, so the reference
itself can not be executed.  The oracle is pinned against the reference's own test
vectors (``tests/golden``): ``tests/case_test.py`` (neighbor ``idx`` exact, targets,
``vel_hist``, ``rel_disp``, ``rel_dist``, ``integrate``) and the Lennard-Jones
"CheatingModel" rollout of ``tests/rollout_test.py`` (MSE < 1e-6).
**The GNS network arithmetic (haiku/jraph) has no golden vector anywhere in the
reference (tests/models_test.py has no GNS case): GNN parity is "parity unpinned"** -
it is a restatement of the published haiku/jraph semantics only.
The third-party neighbor-list algorithm is restated from the published jax-md
source (cell list -> 3^dim stencil candidates -> mask -> cumsum compaction).
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------- utils


class NodeType:
    """lagrangebench/utils.py:17-25."""

    PAD_VALUE = -1
    FLUID = 0
    SOLID_WALL = 1
    MOVING_WALL = 2
    RIGID_BODY = 3
    SIZE = 9


def get_kinematic_mask(particle_type: np.ndarray) -> np.ndarray:
    """lagrangebench/utils.py:28-35."""
    res = np.logical_or(
        particle_type == NodeType.SOLID_WALL, particle_type == NodeType.MOVING_WALL
    )
    return np.logical_or(res, particle_type == NodeType.PAD_VALUE)


# --------------------------------------------------------------------------- space


def _jnp_mod(x: np.ndarray, y) -> np.ndarray:
    """jnp.mod for floats: C fmod, then move the result to the sign of the divisor."""
    r = np.fmod(x, y)
    fix = (r != 0) & ((r < 0) != (np.asarray(y) < 0))
    return np.where(fix, r + y, r)


def space_periodic(side) -> Tuple[Callable, Callable]:
    """jax_md.space.periodic: displacement = mod(dR + side/2, side) - side/2,
    shift = mod(R + dR, side).  Works on (..., dim) arrays (the reference vmaps)."""
    side = np.asarray(side)

    def displacement(a, b):
        dR = a - b
        s = side.astype(dR.dtype)
        return _jnp_mod(dR + s * dR.dtype.type(0.5), s) - dR.dtype.type(0.5) * s

    def shift(r, dr):
        s = side.astype(np.result_type(r, dr))
        return _jnp_mod(r + dr, s)

    return displacement, shift


def space_free() -> Tuple[Callable, Callable]:
    """jax_md.space.free."""
    return (lambda a, b: a - b), (lambda r, dr: r + dr)


def space_distance(dR: np.ndarray) -> np.ndarray:
    """jax_md.space.distance: safe sqrt of the squared norm over the last axis."""
    dr2 = np.sum(dR**2, axis=-1)
    safe = np.where(dr2 > 0, dr2, 1.0)
    return np.where(dr2 > 0, np.sqrt(safe), 0.0).astype(dR.dtype)


# ----------------------------------------------------------------- dataset stats


def get_dataset_stats(metadata: Dict, is_isotropic_norm: bool, noise_std: float, dtype=np.float64):
    """lagrangebench/data/utils.py:9-45."""
    acc_mean = np.array(metadata["acc_mean"], dtype=dtype)
    acc_std = np.array(metadata["acc_std"], dtype=dtype)
    vel_mean = np.array(metadata["vel_mean"], dtype=dtype)
    vel_std = np.array(metadata["vel_std"], dtype=dtype)
    if is_isotropic_norm:
        acc_mean = np.mean(acc_mean) * np.ones_like(acc_mean)
        acc_std = np.sqrt(np.mean(acc_std**2)) * np.ones_like(acc_std)
        vel_mean = np.mean(vel_mean) * np.ones_like(vel_mean)
        vel_std = np.sqrt(np.mean(vel_std**2)) * np.ones_like(vel_std)
    return {
        "acceleration": {"mean": acc_mean, "std": np.sqrt(acc_std**2 + noise_std**2)},
        "velocity": {"mean": vel_mean, "std": np.sqrt(vel_std**2 + noise_std**2)},
    }


# ----------------------------------------------------------------- neighbor list


def _shift_array(arr: np.ndarray, dindex: Sequence[int]) -> np.ndarray:
    """jax_md.partition.shift_array: roll the cell buffer by one cell per axis.
    dindex[k] < 0  ->  arr'[i] = arr[i+1]  (np.roll by -1) along axis k."""
    for axis, d in enumerate(dindex):
        if d < 0:
            arr = np.roll(arr, -1, axis=axis)
        elif d > 0:
            arr = np.roll(arr, 1, axis=axis)
    return arr


@dataclass
class NeighborList:
    """The fields callers of jax_md's NeighborList read (features.py:110,
    rollout.py:135,140) plus the frozen capacities ``update`` re-uses."""

    idx: np.ndarray  # (2, E_cap) int32; row0 receivers, row1 senders; pad = N
    did_buffer_overflow: bool
    cell_capacity: Optional[int]
    max_occupancy: int  # E_cap
    occupancy: int  # number of real edges found (before truncation)
    update_fn: Callable = field(repr=False, default=None)

    def update(self, position, **kwargs):
        return self.update_fn(position, self)


def neighbor_list(
    displacement_fn: Callable,
    box,
    r_cutoff: float,
    capacity_multiplier: float = 1.25,
    mask_self: bool = False,
    dr_threshold: float = 0.0,
):
    """jax_md.partition.neighbor_list, Sparse format, backend "jaxmd_vmap".

    Returns an object with ``allocate(position)`` / ``update(position, nbrs)``.
    Edge ORDER follows the reference: row-major over (scanning particle i =
    "sender", candidate slot); candidates = own cell's ``cap`` slots then the
    3^dim-1 neighbour cells in ``ndindex(3,..)-1`` order.
    """
    box32 = np.asarray(box, dtype=np.float32)  # partition.py casts box to f32
    cutoff = r_cutoff + dr_threshold
    cutoff_sq = cutoff**2  # python double

    def metric_sq(a, b):
        d = displacement_fn(a, b)
        return np.sum(d**2, axis=-1)

    use_cell_list = bool(np.all(np.float32(cutoff) < box32 / np.float32(3.0)))
    dim = box32.size
    if use_cell_list:
        cells_per_side_f = np.floor(box32 / np.float32(cutoff))
        cell_size = (box32 / cells_per_side_f).astype(np.float32)
        cells_per_side = cells_per_side_f.astype(np.int32)
        cell_count = int(np.prod(cells_per_side))
        # x fastest: multipliers = cumprod([1, n_x, n_y, ...])
        hash_mult = np.concatenate([[1], np.cumprod(cells_per_side[:-1])]).astype(np.int64)

    def _hashes(position):
        indices = (position / cell_size.astype(position.dtype)).astype(np.int32)
        return np.sum(indices.astype(np.int64) * hash_mult, axis=1)

    def _candidates_cell(position, cell_capacity):
        N = position.shape[0]
        hashes = _hashes(position)
        max_cell_occ = int(np.bincount(hashes, minlength=cell_count).max())
        order = np.argsort(hashes, kind="stable")
        sorted_hash = hashes[order]
        slot = sorted_hash * cell_capacity + (np.arange(N) % cell_capacity)
        cell_id = np.full((cell_count * cell_capacity,), N, dtype=np.int32)
        ok = slot < cell_id.size
        cell_id[slot[ok]] = order[ok].astype(np.int32)
        grid = tuple(int(c) for c in cells_per_side[::-1])  # (nz, ny, nx)
        buf = cell_id.reshape(grid + (cell_capacity,))
        parts = [buf]
        for dindex in itertools.product(range(3), repeat=dim):
            d = tuple(int(v) - 1 for v in dindex)
            if all(v == 0 for v in d):
                continue
            parts.append(_shift_array(buf, d))
        cand_per_cell = np.concatenate(parts, axis=-1)  # grid + (3^dim * cap,)
        cand_per_cell = cand_per_cell.reshape(cell_count, -1)
        cand = np.full((N + 1, cand_per_cell.shape[1]), 0, dtype=np.int32)
        flat_ids = buf.reshape(-1)
        cell_of_slot = np.repeat(np.arange(cell_count), cell_capacity)
        cand[flat_ids] = cand_per_cell[cell_of_slot]
        return cand[:-1], max_cell_occ

    def _prune(position, cand):
        N = position.shape[0]
        sender = np.broadcast_to(np.arange(N, dtype=np.int32)[:, None], cand.shape).reshape(-1)
        receiver = cand.reshape(-1)
        recv_c = np.minimum(receiver, N - 1)  # JAX clamps OOB gathers
        dR = metric_sq(position[sender], position[recv_c])
        mask = (dR < position.dtype.type(cutoff_sq)) & (receiver < N)
        if mask_self:
            mask &= receiver != sender
        occupancy = int(mask.sum())
        return receiver[mask], sender[mask], occupancy

    def _build(position, nbrs: Optional[NeighborList]):
        position = np.asarray(position)
        N = position.shape[0]
        overflow = False
        if use_cell_list:
            if nbrs is None:
                hashes = _hashes(position)
                cell_capacity = int(
                    np.bincount(hashes, minlength=cell_count).max() * capacity_multiplier
                )
            else:
                cell_capacity = nbrs.cell_capacity
            cand, max_cell_occ = _candidates_cell(position, cell_capacity)
            overflow |= max_cell_occ > cell_capacity
        else:
            cell_capacity = None
            cand = np.broadcast_to(np.arange(N, dtype=np.int32)[None, :], (N, N))
        recv, send, occupancy = _prune(position, cand)
        if nbrs is None:
            max_occupancy = int(occupancy * capacity_multiplier)
            max_occupancy = min(max_occupancy, cand.size)
            limit = N * (N - 1) if mask_self else N * N
            max_occupancy = min(max_occupancy, limit)
        else:
            max_occupancy = nbrs.max_occupancy
        idx = np.full((2, max(max_occupancy, 0)), N, dtype=np.int32)
        k = min(occupancy, max_occupancy)
        idx[0, :k] = recv[:k]
        idx[1, :k] = send[:k]
        overflow |= occupancy > max_occupancy
        return NeighborList(
            idx=idx,
            did_buffer_overflow=bool(overflow),
            cell_capacity=cell_capacity,
            max_occupancy=max_occupancy,
            occupancy=occupancy,
            update_fn=lambda pos, nb: _build(pos, nb),
        )

    class _Fns:
        @staticmethod
        def allocate(position, **kwargs):
            return _build(position, None)

        @staticmethod
        def update(position, nbrs, **kwargs):
            return _build(position, nbrs)

    return _Fns()


def canonical_edges(idx: np.ndarray, n: int) -> np.ndarray:
    """Real edges of a (2, E_cap) list, sorted by (receiver, sender).  This is the
    order the HIP engine emits (CSR by receiver), and the form parity is defined on."""
    real = idx[0] < n
    recv, send = idx[0][real].astype(np.int64), idx[1][real].astype(np.int64)
    order = np.lexsort((send, recv))
    return np.stack([recv[order], send[order]]).astype(np.int32)


# ------------------------------------------------------------------ features


def physical_feature_builder(
    bounds,
    normalization_stats,
    connectivity_radius: float,
    displacement_fn: Callable,
    pbc: Sequence[bool],
    magnitude_features: bool = False,
    external_force_fn: Optional[Callable] = None,
):
    """lagrangebench/case_setup/features.py:13-128."""
    velocity_stats = normalization_stats["velocity"]

    def feature_transform(pos_input: np.ndarray, nbrs: NeighborList) -> Dict[str, np.ndarray]:
        features = {}
        n_total_points = pos_input.shape[0]
        most_recent_position = pos_input[:, -1]
        velocity_sequence = displacement_fn(pos_input[:, 1:], pos_input[:, :-1])
        dt = pos_input.dtype
        normalized_velocity_sequence = (
            velocity_sequence - velocity_stats["mean"].astype(dt)
        ) / velocity_stats["std"].astype(dt)
        features["abs_pos"] = pos_input
        features["vel_hist"] = normalized_velocity_sequence.reshape(n_total_points, -1)
        if magnitude_features:
            features["vel_mag"] = np.linalg.norm(normalized_velocity_sequence, axis=-1)
        if not any(pbc):
            boundaries = np.array(bounds, dtype=dt)
            lo = most_recent_position - boundaries[:, 0][None]
            hi = boundaries[:, 1][None] - most_recent_position
            d2b = np.concatenate([lo, hi], axis=1)
            features["bound"] = np.clip(d2b / dt.type(connectivity_radius), -1.0, 1.0)
        if external_force_fn is not None:
            features["force"] = _vmap_force(external_force_fn, most_recent_position)
        receivers, senders = nbrs.idx
        features["senders"] = senders
        features["receivers"] = receivers
        # JAX clamps the padding index N to N-1 on gather (Appendix A.2 of SURVEY.md)
        rc = np.minimum(receivers, n_total_points - 1)
        sc = np.minimum(senders, n_total_points - 1)
        displacement = displacement_fn(most_recent_position[rc], most_recent_position[sc])
        rel = displacement / dt.type(connectivity_radius)
        features["rel_disp"] = rel
        features["rel_dist"] = space_distance(rel)[:, None]
        return features

    return feature_transform


def _vmap_force(fn, pos):
    """vmap(external_force_fn)(pos): try a vectorised call first, else loop."""
    try:
        out = np.asarray(fn(pos))
        if out.shape == pos.shape:
            return out.astype(pos.dtype)
    except Exception:
        pass
    return np.stack([np.asarray(fn(r)) for r in pos]).astype(pos.dtype)


# ------------------------------------------------------------------ case builder


class CaseSetupFn:
    """lagrangebench/case_setup/case.py:32-59."""

    def __init__(self, allocate, preprocess, allocate_eval, preprocess_eval, integrate,
                 displacement, normalization_stats):
        self.allocate = allocate
        self.preprocess = preprocess
        self.allocate_eval = allocate_eval
        self.preprocess_eval = preprocess_eval
        self.integrate = integrate
        self.displacement = displacement
        self.normalization_stats = normalization_stats


DEFAULT_NEIGHBORS = {"backend": "jaxmd_vmap", "multiplier": 1.25}
DEFAULT_MODEL = {"isotropic_norm": False, "magnitude_features": False}


def case_builder(
    box,
    metadata: Dict,
    input_seq_length: int,
    cfg_neighbors: Optional[Dict] = None,
    cfg_model: Optional[Dict] = None,
    noise_std: float = 3e-4,
    external_force_fn: Optional[Callable] = None,
    dtype=np.float64,
) -> CaseSetupFn:
    """lagrangebench/case_setup/case.py:62-269 (train-mode noise is NOT restated:
    random-walk noise needs the JAX PRNG and is off the inference path)."""
    cfg_neighbors = {**DEFAULT_NEIGHBORS, **(cfg_neighbors or {})}
    cfg_model = {**DEFAULT_MODEL, **(cfg_model or {})}
    dtype = np.dtype(dtype)
    stats = get_dataset_stats(metadata, cfg_model["isotropic_norm"], noise_std, dtype=dtype)

    if np.array(metadata["periodic_boundary_conditions"]).any():
        displacement_fn, shift_fn = space_periodic(np.asarray(box, dtype=dtype))
    else:
        displacement_fn, shift_fn = space_free()

    neighbor_fn = neighbor_list(
        displacement_fn,
        np.asarray(box),
        r_cutoff=metadata["default_connectivity_radius"],
        capacity_multiplier=cfg_neighbors["multiplier"],
        mask_self=False,
    )
    feature_transform = physical_feature_builder(
        bounds=metadata["bounds"],
        normalization_stats=stats,
        connectivity_radius=metadata["default_connectivity_radius"],
        displacement_fn=displacement_fn,
        pbc=metadata["periodic_boundary_conditions"],
        magnitude_features=cfg_model["magnitude_features"],
        external_force_fn=external_force_fn,
    )

    def _compute_target(pos_input):  # case.py:142-160
        cur_v = displacement_fn(pos_input[:, 1], pos_input[:, 0])
        nxt_v = displacement_fn(pos_input[:, 2], pos_input[:, 1])
        acc = nxt_v - cur_v
        a, v = stats["acceleration"], stats["velocity"]
        return {
            "acc": (acc - a["mean"]) / a["std"],
            "vel": (nxt_v - v["mean"]) / v["std"],
            "pos": pos_input[:, -1],
        }

    def _preprocess(sample, neighbors=None, is_allocate=False, mode="train", **kw):
        pos_input = np.asarray(sample[0], dtype=dtype)
        if mode == "train":
            if kw.get("noise_std", 0.0) != 0.0:
                raise NotImplementedError("random-walk noise (train/strats.py) is not restated")
            unroll_steps = kw.get("unroll_steps", 0)
        most_recent_position = pos_input[:, input_seq_length - 1]
        if is_allocate:
            neighbors = neighbor_fn.allocate(most_recent_position)
        else:
            neighbors = neighbors.update(most_recent_position)
        features = feature_transform(pos_input[:, :input_seq_length], neighbors)
        if mode == "train":
            b = input_seq_length - 2 + unroll_steps
            target = _compute_target(pos_input[:, b : b + 3])
            return kw.get("key"), features, target, neighbors
        return features, neighbors

    def allocate_fn(key, sample, noise_std=0.0, unroll_steps=0):
        return _preprocess(sample, key=key, noise_std=noise_std, unroll_steps=unroll_steps,
                           is_allocate=True)

    def preprocess_fn(key, sample, noise_std, neighbors, unroll_steps=0):
        return _preprocess(sample, neighbors, key=key, noise_std=noise_std,
                           unroll_steps=unroll_steps)

    def allocate_eval_fn(sample):
        return _preprocess(sample, is_allocate=True, mode="eval")

    def preprocess_eval_fn(sample, neighbors):
        return _preprocess(sample, neighbors, mode="eval")

    def integrate_fn(normalized_in, position_sequence):  # case.py:230-259
        if "pos" in normalized_in:
            return normalized_in["pos"]
        most_recent_position = position_sequence[:, -1]
        if "vel" in normalized_in:
            v = stats["velocity"]
            new_velocity = v["mean"] + normalized_in["vel"] * v["std"]
        else:
            a = stats["acceleration"]
            acceleration = a["mean"] + normalized_in["acc"] * a["std"]
            most_recent_velocity = displacement_fn(most_recent_position, position_sequence[:, -2])
            new_velocity = most_recent_velocity + acceleration
        return shift_fn(most_recent_position, new_velocity)

    return CaseSetupFn(allocate_fn, preprocess_fn, allocate_eval_fn, preprocess_eval_fn,
                       integrate_fn, displacement_fn, stats)


# ---------------------------------------------------------------------- GNS


def _trunc_normal(rng: np.random.Generator, shape, stddev: float) -> np.ndarray:
    """hk.initializers.TruncatedNormal: N(0,1) truncated to [-2, 2], times stddev."""
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


def gns_layer_names(num_mp_steps: int) -> List[str]:
    """Creation order of the MLP+LayerNorm blocks (gns.py:65-133): encoder node,
    encoder edge, then per MP step edge fn and node fn, then the decoder (no LN)."""
    names = ["enc_node", "enc_edge"]
    for k in range(num_mp_steps):
        names += [f"proc{k}_edge", f"proc{k}_node"]
    names.append("decoder")
    return names


def gns_init(
    rng: np.random.Generator,
    node_in: int,
    edge_in: int,
    particle_dimension: int,
    latent_size: int = 128,
    blocks_per_step: int = 2,
    num_mp_steps: int = 10,
    particle_type_embedding_size: int = 16,
    num_particle_types: int = NodeType.SIZE,
    decoder_scale: float = 1.0,
) -> Dict[str, Dict[str, np.ndarray]]:
    """Haiku-default init: Linear w ~ TruncNormal(1/sqrt(fan_in)), b=0;
    LayerNorm scale=1, offset=0; Embed ~ TruncNormal(1).  ``node_in`` excludes the
    embedding, which is concatenated when num_particle_types > 1 (gns.py:164-169)."""
    p: Dict[str, Dict[str, np.ndarray]] = {}
    if num_particle_types > 1:
        p["embed"] = {"embeddings": _trunc_normal(rng, (num_particle_types, particle_type_embedding_size), 1.0)}
        node_in = node_in + particle_type_embedding_size

    def mlp(name, fan_in, out, layer_norm=True, scale=1.0):
        sizes = [latent_size] * (blocks_per_step - 1) + [out]
        d = fan_in
        for li, s in enumerate(sizes):
            w = _trunc_normal(rng, (d, s), 1.0 / np.sqrt(d))
            if li == len(sizes) - 1:
                w = (w * scale).astype(np.float32)
            p[f"{name}/linear_{li}"] = {"w": w, "b": np.zeros((s,), np.float32)}
            d = s
        if layer_norm:
            p[f"{name}/layer_norm"] = {"scale": np.ones((out,), np.float32),
                                       "offset": np.zeros((out,), np.float32)}

    L = latent_size
    mlp("enc_node", node_in, L)
    mlp("enc_edge", edge_in, L)
    for k in range(num_mp_steps):
        mlp(f"proc{k}_edge", 3 * L, L)
        mlp(f"proc{k}_node", 2 * L, L)
    mlp("decoder", L, particle_dimension, layer_norm=False, scale=decoder_scale)
    return p


def _mlp_apply(p, name, x, blocks_per_step):
    """hk.nets.MLP (ReLU, activate_final=False) [+ hk.LayerNorm(axis=-1, eps=1e-5)]."""
    for li in range(blocks_per_step):
        lin = p[f"{name}/linear_{li}"]
        x = x @ lin["w"] + lin["b"]
        if li < blocks_per_step - 1:
            x = np.maximum(x, np.float32(0))
    ln = p.get(f"{name}/layer_norm")
    if ln is not None:
        mean = x.mean(axis=-1, keepdims=True, dtype=np.float32)
        var = np.mean(np.square(x - mean), axis=-1, keepdims=True, dtype=np.float32)
        inv = ln["scale"] * (np.float32(1) / np.sqrt(var + np.float32(1e-5)))
        x = inv * (x - mean) + ln["offset"]
    return x.astype(np.float32)


def segment_sum(data: np.ndarray, segment_ids: np.ndarray, num_segments: int) -> np.ndarray:
    """jraph.segment_sum: ids >= num_segments are dropped."""
    out = np.zeros((num_segments,) + data.shape[1:], dtype=data.dtype)
    ok = segment_ids < num_segments
    np.add.at(out, segment_ids[ok], data[ok])
    return out


def gns_transform(features: Dict[str, np.ndarray]):
    """gns.py:135-157: node = [vel_hist, vel_mag?, bound?, force?]; edge = [rel_disp, rel_dist]."""
    nodes = np.concatenate(
        [features[k].reshape(features["vel_hist"].shape[0], -1)
         for k in ["vel_hist", "vel_mag", "bound", "force"] if k in features], axis=-1)
    edges = np.concatenate([features[k] for k in ["rel_disp", "rel_dist"] if k in features], axis=-1)
    return nodes.astype(np.float32), edges.astype(np.float32)


def gns_apply(
    params,
    features: Dict[str, np.ndarray],
    particle_type: np.ndarray,
    num_mp_steps: int = 10,
    blocks_per_step: int = 2,
    skip_padding: bool = False,
    return_intermediates: bool = False,
):
    """GNS.__call__ (gns.py:159-171), all-fp32 (runner.py:71-72 jmp policy).

    ``skip_padding=False`` keeps the reference's algorithmic shape: the MLPs run over
    all E_cap rows; padded rows gather node N-1 (JAX clamp) and are dropped by
    segment_sum.  ``skip_padding=True`` evaluates only real edges (same node outputs).
    """
    nodes, edges = gns_transform(features)
    n_nodes = nodes.shape[0]
    senders = features["senders"].astype(np.int64)
    receivers = features["receivers"].astype(np.int64)
    if skip_padding:
        real = receivers < n_nodes
        senders, receivers, edges = senders[real], receivers[real], edges[real]
    if "embed" in params:
        pt = np.where(particle_type < 0, particle_type + NodeType.SIZE, particle_type)
        nodes = np.concatenate([nodes, params["embed"]["embeddings"][pt]], axis=-1)
    inter = {}
    n = _mlp_apply(params, "enc_node", nodes, blocks_per_step)
    e = _mlp_apply(params, "enc_edge", edges, blocks_per_step)
    sc = np.minimum(senders, n_nodes - 1)
    rc = np.minimum(receivers, n_nodes - 1)
    if return_intermediates:
        inter["enc_n"], inter["enc_e"] = n.copy(), e.copy()
    for k in range(num_mp_steps):
        ein = np.concatenate([n[sc], n[rc], e], axis=-1)
        e2 = _mlp_apply(params, f"proc{k}_edge", ein, blocks_per_step)
        agg = segment_sum(e2, receivers, n_nodes)
        n2 = _mlp_apply(params, f"proc{k}_node", np.concatenate([n, agg], axis=-1), blocks_per_step)
        n = n2 + n
        e = e2 + e
        if return_intermediates:
            inter[f"n{k}"], inter[f"agg{k}"] = n.copy(), agg.copy()
    acc = _mlp_apply(params, "decoder", n, blocks_per_step)
    if return_intermediates:
        return {"acc": acc}, inter
    return {"acc": acc}


# ------------------------------------------------------------------- rollout


def forward_eval(model_apply, params, state, sample, current_positions, target_positions,
                 case_integrate):
    """lagrangebench/evaluate/rollout.py:31-75."""
    _, particle_type = sample
    pred, state = model_apply(params, state, sample)
    next_position = case_integrate(pred, current_positions)
    kinematic_mask = get_kinematic_mask(particle_type)
    next_position = np.where(kinematic_mask[:, None], target_positions, next_position)
    current_positions = np.concatenate(
        [current_positions[:, 1:], next_position[:, None, :]], axis=1)
    return current_positions, state


def mse(displacement_fn, pred, target):
    """metrics.py:139-142 for one frame."""
    return float((displacement_fn(pred, target) ** 2).mean())


def mae(displacement_fn, pred, target):
    """metrics.py:144-147 for one frame."""
    return float(np.abs(displacement_fn(pred, target)).mean())


def e_kin(displacement_fn, rollout, stride, dt, dx, dim):
    """metrics.py:98-125,157-160 for one rollout (T, N, dim): dx^dim * sum((v/dt)^2) per strided frame."""
    v = displacement_fn(rollout[1::stride], rollout[0:-1:stride])
    return ((v / dt) ** 2).sum(axis=(1, 2)) * dx**dim


def metrics_mse_mae(displacement_fn, pred_rollout, target_rollout, active=("mse",),
                    loss_ranges=(1, 5, 10, 20, 50, 100)):
    """metrics.py:69-96: per-step metric and the shorter-horizon slices."""
    target_rollout = np.asarray(target_rollout, dtype=pred_rollout.dtype)
    out = {}
    for name in active:
        fn = {"mse": mse, "mae": mae}[name]
        v = np.array([fn(displacement_fn, p, t) for p, t in zip(pred_rollout, target_rollout)])
        out[name] = v
        for i in loss_ranges:
            if i < v.shape[0]:
                out[f"{name}{i}"] = v[:i]
    return out


def eval_batched_rollout(model_apply, case: CaseSetupFn, params, state, traj_batch_i,
                         neighbors: NeighborList, n_rollout_steps: int, t_window: int,
                         n_extrap_steps: int = 0, metrics=("mse",), verbose=False):
    """lagrangebench/evaluate/rollout.py:78-178, one trajectory at a time (the
    reference vmaps over the batch; trajectories never interact)."""
    pos_input_batch, particle_type_batch = traj_batch_i
    B, n_nodes, _, dim = pos_input_batch.shape
    if n_rollout_steps == -1:
        n_rollout_steps = pos_input_batch.shape[2] - t_window
    traj_len = n_rollout_steps + n_extrap_steps
    predictions = np.zeros((B, traj_len, n_nodes, dim), dtype=np.float64)
    metrics_batch = []
    n_realloc = 0
    for b in range(B):
        cur = np.asarray(pos_input_batch[b][:, 0:t_window], dtype=np.float64)
        target = pos_input_batch[b][:, t_window : t_window + traj_len]
        ptype = particle_type_batch[b]
        nbrs = neighbors
        st = state
        step = 0
        while step < traj_len:
            feats, nbrs = case.preprocess_eval((cur, ptype), nbrs)
            if nbrs.did_buffer_overflow:
                if verbose:
                    print(f"(eval) Reallocate neighbors list at step {step}")
                _, nbrs = case.allocate_eval((cur, ptype))
                n_realloc += 1
                continue
            tstep = min(step, target.shape[1] - 1)  # JAX clamps the OOB gather
            cur, st = forward_eval(model_apply, params, st, (feats, ptype), cur,
                                   target[:, tstep], case.integrate)
            predictions[b, step] = cur[:, -1]
            step += 1
        tgt = np.transpose(target, (1, 0, 2))
        metrics_batch.append(
            metrics_mse_mae(case.displacement, predictions[b, :n_rollout_steps], tgt[:n_rollout_steps],
                            active=metrics))
        neighbors_out = nbrs if b == 0 else neighbors_out
    return predictions, metrics_batch, neighbors_out
