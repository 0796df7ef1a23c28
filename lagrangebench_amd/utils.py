"""General utils mirrored from lagrangebench/utils.py (reference :17-47)."""
from __future__ import annotations

import enum
from typing import Any

import numpy as np
import torch


class NodeType(enum.IntEnum):
    """Particle types - lagrangebench/utils.py:17-25."""

    PAD_VALUE = -1
    FLUID = 0
    SOLID_WALL = 1
    MOVING_WALL = 2
    RIGID_BODY = 3
    SIZE = 9


def get_kinematic_mask(particle_type):
    """True for kinematic (obstacle / padding) particles - lagrangebench/utils.py:28-35.
    (Inside the engine the same rule is applied by k_integrate.)"""
    if isinstance(particle_type, torch.Tensor):
        return (particle_type == NodeType.SOLID_WALL) | (particle_type == NodeType.MOVING_WALL) | (
            particle_type == NodeType.PAD_VALUE)
    pt = np.asarray(particle_type)
    return (pt == NodeType.SOLID_WALL) | (pt == NodeType.MOVING_WALL) | (pt == NodeType.PAD_VALUE)


def _tree_map(fn, tree: Any):
    if hasattr(tree, "_lb_tree_map"):
        return tree._lb_tree_map(fn)
    if isinstance(tree, dict):
        return type(tree)((k, _tree_map(fn, v)) for k, v in tree.items())
    if isinstance(tree, (tuple, list)):
        return type(tree)(_tree_map(fn, v) for v in tree)
    if tree is None:
        return None
    return fn(tree)


def broadcast_to_batch(sample, batch_size: int):
    """Broadcast a pytree to a batched one with first dimension batch_size (utils.py:38-41)."""
    assert batch_size > 0

    def rep(x):
        if isinstance(x, torch.Tensor):
            return x[None].expand(batch_size, *x.shape).clone()
        x = np.asarray(x)
        return np.repeat(x[None, ...], batch_size, axis=0)

    return _tree_map(rep, sample)


def broadcast_from_batch(batch, index: int):
    """Pick sample `index` out of a batched pytree (utils.py:44-47)."""
    assert index >= 0
    return _tree_map(lambda x: x[index], batch)


# ---------------------------------------------------------------------------------------------
# Checkpoints - mirror of lagrangebench/utils.py:50-128.
#
# On-disk format of the reference (save_pytree): `<name>_array.npy` = the leaves of the pytree,
# `np.save`d back to back in `jax.tree_leaves` order (dict keys sorted at every level);
# `<name>_tree.pkl` = the same nested dict with every leaf replaced by 0 (plain pickle: dict / str /
# int only, so it loads without JAX); `metadata_ckp.json` = {"step", "loss"}; `opt_state.pkl`
# (cloudpickle of the optax state) is training-only and is neither read nor written here.
import json as _json
import os as _os
import pickle as _pickle
import re as _re


def _tree_leaves_sorted(tree, prefix=()):
    """[(path, leaf)] in jax.tree_leaves order for nested dicts."""
    if isinstance(tree, dict):
        out = []
        for k in sorted(tree):
            out += _tree_leaves_sorted(tree[k], prefix + (k,))
        return out
    return [(prefix, tree)]


def _tree_set(tree, path, value):
    for k in path[:-1]:
        tree = tree[k]
    tree[path[-1]] = value


def save_pytree(ckp_dir: str, pytree_obj, name: str) -> None:
    """utils.py:50-58."""
    with open(_os.path.join(ckp_dir, f"{name}_array.npy"), "wb") as f:
        for _, x in _tree_leaves_sorted(pytree_obj):
            np.save(f, np.asarray(x), allow_pickle=False)

    def zeros(t):
        return {k: zeros(v) for k, v in t.items()} if isinstance(t, dict) else 0
    with open(_os.path.join(ckp_dir, f"{name}_tree.pkl"), "wb") as f:
        _pickle.dump(zeros(pytree_obj), f)


def load_pytree(model_dir: str, name: str):
    """utils.py:99-109."""
    with open(_os.path.join(model_dir, f"{name}_tree.pkl"), "rb") as f:
        tree = _pickle.load(f)
    leaves = _tree_leaves_sorted(tree)
    with open(_os.path.join(model_dir, f"{name}_array.npy"), "rb") as f:
        for path, _ in leaves:
            arr = np.load(f)
            if path:
                _tree_set(tree, path, arr)
            else:
                tree = arr
    return tree


def _opt_state_to_host(x):
    """torch tensors -> numpy (recursively) so that opt_state.pkl loads without a device."""
    try:
        import torch
        if isinstance(x, torch.Tensor):
            return x.detach().cpu().numpy()
    except ImportError:  # pragma: no cover
        pass
    if isinstance(x, dict):
        return {k: _opt_state_to_host(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_opt_state_to_host(v) for v in x)
    return x


def save_haiku(ckp_dir: str, params, state, opt_state, metadata_ckp) -> None:
    """utils.py:61-96 incl. the best/ copy.  `opt_state.pkl` is always written (the reference's load_haiku opens
    it unconditionally, utils.py:119-121): here it holds the device AdamW state of train/trainer.py as numpy arrays -
    {"kind", "m", "v": the two moment blobs (flat, GNS.flatten order), "step": the training-loop index, "count": the
    AdamW steps taken (optax's count: the bias-correction exponent)} - or None.  It is NOT an optax state: a reference
    run can read params / state of these checkpoints, but restarts its optimiser from scratch."""
    _os.makedirs(ckp_dir, exist_ok=True)
    save_pytree(ckp_dir, params, "params")
    save_pytree(ckp_dir, state if state is not None else {}, "state")
    with open(_os.path.join(ckp_dir, "opt_state.pkl"), "wb") as f:
        _pickle.dump(_opt_state_to_host(opt_state), f)
    with open(_os.path.join(ckp_dir, "metadata_ckp.json"), "w") as f:
        _json.dump(metadata_ckp, f)
    if "best" not in ckp_dir:
        best = _os.path.join(ckp_dir, "best")
        meta_best = _os.path.join(best, "metadata_ckp.json")
        if _os.path.exists(meta_best):
            with open(meta_best) as fp:
                prev = _json.loads(fp.read())
            if metadata_ckp["loss"] < prev["loss"]:
                save_haiku(best, params, state, opt_state, metadata_ckp)
        else:
            save_haiku(best, params, state, opt_state, metadata_ckp)


def load_haiku(model_dir: str):
    """utils.py:112-128: (params, state, opt_state, step).  opt_state is whatever opt_state.pkl holds when it
    unpickles without JAX / optax (this package's AdamW state_dict, or None); an optax state written by the
    reference needs cloudpickle + optax and is returned as None."""
    params = load_pytree(model_dir, "params")
    state = load_pytree(model_dir, "state") if _os.path.exists(_os.path.join(model_dir, "state_tree.pkl")) else {}
    opt_state = None
    opt_path = _os.path.join(model_dir, "opt_state.pkl")
    if _os.path.exists(opt_path):
        try:
            with open(opt_path, "rb") as f:
                opt_state = _pickle.load(f)
        except Exception:  # an optax state of the reference: not loadable without JAX
            opt_state = None
    step = 0
    meta = _os.path.join(model_dir, "metadata_ckp.json")
    if _os.path.exists(meta):
        with open(meta) as fp:
            step = _json.loads(fp.read()).get("step", 0)
    print(f"Loaded model from {model_dir} at step {step}")
    return params, state, opt_state, step


_SCOPE_RANK = (("encoder", 0), ("processor", 1), ("decoder", 2))


def gns_params_from_haiku(hk_params, num_mp_steps: int, blocks_per_step: int = 2):
    """Map a Haiku GNS parameter dict {module_name: {"w","b"} | {"scale","offset"} | {"embeddings"}}
    onto this package's layout ("embed", "<block>/linear_i", "<block>/layer_norm", blocks in module
    creation order enc_node, enc_edge, (proc_k_edge, proc_k_node)*, decoder - models/gns.py:65-133).

    Haiku uniquifies module names with numeric suffixes in creation order (MLP, MLP_1, ...,
    layer_norm, layer_norm_1, ...), possibly per method scope (~_encoder / ~_processor / ~_decoder);
    modules are therefore ordered by (scope rank, suffix) - exact prefixes do not matter."""
    from .models.gns import layer_names
    names = layer_names(num_mp_steps)

    def order_key(mod: str, stem: str):
        m = _re.search(rf"(?:^|/)({stem})(?:_(\d+))?(?:/|$)", mod)
        idx = int(m.group(2)) if m and m.group(2) else 0
        scope = mod[: m.start()] if m else mod
        rank = next((r for s, r in _SCOPE_RANK if s in scope), 1)
        return (rank, idx)

    lin, lns, embed = {}, [], None
    for mod, leaves in hk_params.items():
        if "embeddings" in leaves:
            embed = leaves["embeddings"]
        elif "scale" in leaves and "offset" in leaves:
            lns.append((order_key(mod, "layer_norm"), leaves))
        elif "w" in leaves:
            m = _re.search(r"linear(?:_(\d+))?$", mod)
            li = int(m.group(1)) if m and m.group(1) else 0
            lin.setdefault(order_key(mod, "MLP"), {})[li] = leaves
    mlps = [lin[k] for k in sorted(lin)]
    lns = [v for _, v in sorted(lns, key=lambda kv: kv[0])]
    if len(mlps) != len(names) or len(lns) != len(names) - 1:
        raise ValueError(f"haiku params hold {len(mlps)} MLPs / {len(lns)} LayerNorms, expected "
                         f"{len(names)} / {len(names) - 1} for num_mp_steps={num_mp_steps}")
    out = {}
    if embed is not None:
        out["embed"] = {"embeddings": np.asarray(embed, np.float32)}
    for i, name in enumerate(names):
        for li in range(blocks_per_step):
            leaf = mlps[i][li]
            out[f"{name}/linear_{li}"] = {"w": np.asarray(leaf["w"], np.float32),
                                         "b": np.asarray(leaf["b"], np.float32)}
        if i < len(lns):
            out[f"{name}/layer_norm"] = {"scale": np.asarray(lns[i]["scale"], np.float32),
                                         "offset": np.asarray(lns[i]["offset"], np.float32)}
    return out


def gns_params_to_haiku(params, num_mp_steps: int, blocks_per_step: int = 2, module: str = "gns"):
    """Inverse of gns_params_from_haiku with flat creation-order suffixes."""
    from .models.gns import layer_names
    out = {}
    if "embed" in params:
        out[f"{module}/~/embed"] = {"embeddings": params["embed"]["embeddings"]}
    for i, name in enumerate(layer_names(num_mp_steps)):
        sfx = "" if i == 0 else f"_{i}"
        for li in range(blocks_per_step):
            out[f"{module}/MLP{sfx}/~/linear_{li}"] = dict(params[f"{name}/linear_{li}"])
        if f"{name}/layer_norm" in params:
            out[f"{module}/layer_norm{sfx}"] = dict(params[f"{name}/layer_norm"])
    return out


# ---------------------------------------------------------------------------------------------
# SEGNN checkpoints (models/segnn.py:30-128,252-362 + e3nn-jax 0.20.3's haiku Linear).
#
# Every O3TensorProduct of the reference owns ONE e3nn.haiku.Linear whose parameters are named
# "w[i_in,i_out] <mul>x<ir_in>,<mul>x<ir_out>" (shape (mul_in, mul_out)) and "b[i_out] <mul>x0e"
# ([mem] - e3nn-jax is not installable here; tests/golden/make_jax_golden.py dumps real names so that
# tests/test_jax_golden.py can confirm this the day a JAX machine is available).  For lmax 1 the
# engine needs, per block, ws = the 0e->0e matrix, wv = the 1o->1o matrix and b.
#
# Row order.  e3nn.tensor_product(x, y) emits one chunk per (x chunk, y chunk, ir_out) in loop order
# and regroups them with a STABLE sort by irrep, so the rows of the 0e (and of the 1o) matrix follow the
# chunk order of x: a "mul x 0e" chunk contributes its `mul` scalar-derived rows, a "mul x 1o" chunk its
# `mul` vector-derived rows, in the order the chunks appear.  The engine (oracle/segnn_oracle.py:
# tp_inputs) orders every OPERAND as [all scalar channels | all vector channels].  The two agree for
# "32x0e+32x1o" operands and differ for the node features ("5x1o+1x1o+9x0e": vectors first) and for the
# additional message features ("1x1o+1x0e"): `segnn_row_order` is that permutation.
_W_RE = _re.compile(r"^w\[(\d+),(\d+)\]\s+(\d+)x(\d)([eo]),(\d+)x(\d)([eo])$")
_B_RE = _re.compile(r"^b\[(\d+)\]\s+(\d+)x0e$")


def _parse_chunks(irreps: str):
    """"5x1o+1x1o+9x0e" -> [(5, 1), (1, 1), (9, 0)]."""
    out = []
    for term in str(irreps).replace(" ", "").split("+"):
        m = _re.match(r"^(\d+)x(\d)([eo])$", term)
        if not m:
            raise ValueError(f"cannot parse irreps term {term!r}")
        out.append((int(m.group(1)), int(m.group(2))))
    return out


def segnn_row_order(operands) -> np.ndarray:
    """perm with engine_rows = e3nn_rows[perm] for a tensor-product input made of `operands`
    (each a list of (mul, l) chunks in e3nn order)."""
    e3nn_pos, pos = [], 0
    for chunks in operands:          # position of every channel in e3nn's row order
        per = []
        for mul, l in chunks:
            per.append((l, list(range(pos, pos + mul))))
            pos += mul
        e3nn_pos.append(per)
    perm = []
    for per in e3nn_pos:             # engine order: per operand, scalar channels then vector channels
        for want in (0, 1):
            for l, idx in per:
                if l == want:
                    perm += idx
    return np.asarray(perm, dtype=np.int64)


def _segnn_block_operands(model):
    """Operand chunk lists of every O3TensorProduct of `model` (a lagrangebench_amd.models.SEGNN) in the
    engine's block order (SEGNN.block_shapes)."""
    C, B = model._hidden, model._blocks_per_step
    hid = [(C, 0), (C, 1)]
    ops = [[_parse_chunks(model._node_irreps_str)]]
    for _ in range(model._num_mp_steps):
        for i in range(B):
            ops.append([hid, hid, [(1, 1), (1, 0)]] if i == 0 else [hid])     # "1x1o+1x0e" (runner.py:227)
        for i in range(B):
            ops.append([hid, hid] if i == 0 else [hid])
    for _ in range(B):
        ops.append([hid])
    ops.append([hid])
    return ops


def _segnn_module_order(hk_params, num_mp_steps: int, blocks_per_step: int):
    """Haiku module names -> the engine's block order.  Inside `layer_k` the message blocks are created
    first (tp_0 .. tp_{B-1}), then the update blocks reuse the names and get Haiku's numeric suffix
    (tp_0_1 ..): order by (suffix, index)."""
    def find(pred, what):
        hits = [k for k in hk_params if pred(k)]
        if len(hits) != 1:
            raise ValueError(f"SEGNN checkpoint: {len(hits)} modules match {what}: {hits}")
        return hits[0]
    order = [find(lambda k: "embedding_nodes" in k, "embedding_nodes")]
    for n in range(num_mp_steps):
        mods = []
        for k in hk_params:
            m = _re.search(rf"(?:^|/)layer_{n}(?:/|$).*?tp_(\d+)(?:_(\d+))?(?:/|$)", k)
            if m:
                mods.append(((int(m.group(2) or 0), int(m.group(1))), k))
        mods.sort()
        if len(mods) != 2 * blocks_per_step:
            raise ValueError(f"SEGNN checkpoint: layer_{n} holds {len(mods)} tensor products, expected {2 * blocks_per_step}")
        order += [k for _, k in mods]
    for i in range(blocks_per_step):
        order.append(find(lambda k, i=i: _re.search(rf"readout_{i}(?:/|$)", k) is not None, f"readout_{i}"))
    order.append(find(lambda k: _re.search(r"(?:^|/)output(?:/|$)", k) is not None, "output"))
    return order


# General irreps / norm (models.SEGNN.generic): the engine keeps e3nn's own row order, one matrix per output irrep
# ("w{l}"), so the leaves map one to one.  e3nn.haiku.BatchNorm modules ([mem]: "batch_norm", "batch_norm_1" inside
# layer_k in creation order - messages first for "batch" - with parameters "weight" / "bias"; the running statistics
# live in Haiku STATE and are never read: the reference calls the module in training mode, segnn.py:303,347-351).
_IR = {0: "0e", 1: "1o", 2: "2e"}


def _segnn_bn_modules(hk_params, n: int):
    mods = []
    for k in hk_params:
        m = _re.search(rf"(?:^|/)layer_{n}(?:/|$).*?batch_norm(?:_(\d+))?(?:/|$)", k)
        if m:
            mods.append((int(m.group(1) or 0), k))
    return [k for _, k in sorted(mods)]


def _segnn_generic_from_haiku(hk_params, model):
    mods = _segnn_module_order(hk_params, model._num_mp_steps, model._blocks_per_step)
    out = {}
    for (name, _, _), mod in zip(model.gen_blocks(), mods):
        blk = {}
        for leaf, arr in hk_params[mod].items():
            m = _W_RE.match(leaf)
            if m:
                if int(m.group(4)) != int(m.group(7)):
                    raise ValueError(f"{mod}/{leaf}: path between different irreps")
                blk[f"w{int(m.group(4))}"] = np.asarray(arr, np.float32)
            elif _B_RE.match(leaf):
                blk["b"] = np.asarray(arr, np.float32)
        out[name] = blk
    which = {0: [], 1: ["norm_nodes"], 2: ["norm_msg", "norm_nodes"]}[model._norm]
    for n in range(model._num_mp_steps if which else 0):
        bn = _segnn_bn_modules(hk_params, n)
        if len(bn) != len(which):
            raise ValueError(f"SEGNN checkpoint: layer_{n} holds {len(bn)} BatchNorm modules, expected {len(which)}")
        for w, mod in zip(which, bn):
            out[f"layer_{n}/{w}"] = {"weight": np.asarray(hk_params[mod]["weight"], np.float32).ravel(),
                                    "bias": np.asarray(hk_params[mod]["bias"], np.float32).ravel()}
    for blk, leaf, shape in model.gen_leaves():     # unreachable / absent leaves stay zero
        out.setdefault(blk, {}).setdefault(leaf, np.zeros(shape, np.float32))
        if out[blk][leaf].shape != shape:
            raise ValueError(f"{blk}/{leaf}: got {out[blk][leaf].shape}, expected {shape}")
    return out


def _segnn_generic_to_haiku(params, model, module: str = "segnn"):
    B = model._blocks_per_step
    names = ["o3_embedding/embedding_nodes"]
    for n in range(model._num_mp_steps):
        names += [f"layer_{n}/tp_{i}" for i in range(B)] + [f"layer_{n}/tp_{i}_1" for i in range(B)]
    names += [f"o3_decoder/readout_{i}" for i in range(B)] + ["o3_decoder/output"]
    out = {}
    for (name, _, outs), hk_name in zip(model.gen_blocks(), names):
        leaves, blk = {}, params[name]
        present = [(mul, l) for mul, l in outs if f"w{l}" in blk]
        for i, (mul, l) in enumerate(present):
            w = np.asarray(blk[f"w{l}"], np.float32)
            leaves[f"w[{i},{i}] {w.shape[0]}x{_IR[l]},{mul}x{_IR[l]}"] = w
        if "b" in blk and np.asarray(blk["b"]).size:
            leaves[f"b[0] {np.asarray(blk['b']).size}x0e"] = np.asarray(blk["b"], np.float32)
        out[f"{module}/{hk_name}/linear"] = leaves
    which = {0: [], 1: ["norm_nodes"], 2: ["norm_msg", "norm_nodes"]}[model._norm]
    for n in range(model._num_mp_steps if which else 0):
        for j, w in enumerate(which):
            out[f"{module}/layer_{n}/batch_norm" + (f"_{j}" if j else "")] = {
                "weight": np.asarray(params[f"layer_{n}/{w}"]["weight"], np.float32),
                "bias": np.asarray(params[f"layer_{n}/{w}"]["bias"], np.float32)}
    return out


def segnn_params_from_haiku(hk_params, model):
    """Haiku/e3nn SEGNN parameter dict -> this package's {block: {"ws", "wv", "b"}} layout
    (lagrangebench_amd.models.SEGNN.block_shapes).  `model`: the SEGNN instance (its irreps fix the
    row permutation)."""
    if getattr(model, "generic", False):
        return _segnn_generic_from_haiku(hk_params, model)
    shapes = model.block_shapes()
    mods = _segnn_module_order(hk_params, model._num_mp_steps, model._blocks_per_step)
    operands = _segnn_block_operands(model)
    out = {}
    for (name, K, ms, mv), mod, ops in zip(shapes, mods, operands):
        ws = wv = b = None
        for leaf, arr in hk_params[mod].items():
            m = _W_RE.match(leaf)
            if m:
                l_in, l_out = int(m.group(4)), int(m.group(7))
                if l_in != l_out:
                    raise ValueError(f"{mod}/{leaf}: path between different irreps")
                if l_in == 0:
                    ws = np.asarray(arr, np.float32)
                elif l_in == 1 and m.group(5) == "o":
                    wv = np.asarray(arr, np.float32)
                continue
            if _B_RE.match(leaf):
                b = np.asarray(arr, np.float32)
        perm = segnn_row_order(ops)
        if len(perm) != K:
            raise ValueError(f"{name}: {len(perm)} tensor-product channels, expected {K}")
        ws = np.zeros((K, ms), np.float32) if ws is None else ws[perm]
        wv = np.zeros((K, mv), np.float32) if wv is None else wv[perm]
        b = np.zeros((ms,), np.float32) if b is None else b
        if ws.shape != (K, ms) or wv.shape != (K, mv) or b.shape != (ms,):
            raise ValueError(f"{name} <- {mod}: got ws {ws.shape} wv {wv.shape} b {b.shape}, expected {(K, ms)} {(K, mv)} {(ms,)}")
        out[name] = {"ws": ws, "wv": wv, "b": b}
    return out


def segnn_params_to_haiku(params, model, module: str = "segnn"):
    """Inverse of segnn_params_from_haiku (e3nn leaf names, e3nn row order)."""
    if getattr(model, "generic", False):
        return _segnn_generic_to_haiku(params, model, module)
    shapes = model.block_shapes()
    operands = _segnn_block_operands(model)
    B = model._blocks_per_step
    names = ["o3_embedding/embedding_nodes"]
    for n in range(model._num_mp_steps):
        names += [f"layer_{n}/tp_{i}" for i in range(B)] + [f"layer_{n}/tp_{i}_1" for i in range(B)]
    names += [f"o3_decoder/readout_{i}" for i in range(B)] + ["o3_decoder/output"]
    out = {}
    for (name, K, ms, mv), hk_name, ops in zip(shapes, names, operands):
        inv = np.argsort(segnn_row_order(ops))
        blk = params[name]
        leaves = {}
        if ms:
            leaves[f"w[0,0] {K}x0e,{ms}x0e"] = np.asarray(blk["ws"], np.float32)[inv]
            leaves[f"b[0] {ms}x0e"] = np.asarray(blk["b"], np.float32)
        leaves[f"w[1,{1 if ms else 0}] {K}x1o,{mv}x1o"] = np.asarray(blk["wv"], np.float32)[inv]
        out[f"{module}/{hk_name}/linear"] = leaves
    return out
