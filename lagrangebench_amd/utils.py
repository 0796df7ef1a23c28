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


def save_haiku(ckp_dir: str, params, state, opt_state, metadata_ckp) -> None:
    """utils.py:61-96 (opt_state is ignored: training is out of scope), incl. the best/ copy."""
    _os.makedirs(ckp_dir, exist_ok=True)
    save_pytree(ckp_dir, params, "params")
    save_pytree(ckp_dir, state if state is not None else {}, "state")
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
    """utils.py:112-128: (params, state, opt_state, step); opt_state is returned as None."""
    params = load_pytree(model_dir, "params")
    state = load_pytree(model_dir, "state") if _os.path.exists(_os.path.join(model_dir, "state_tree.pkl")) else {}
    step = 0
    meta = _os.path.join(model_dir, "metadata_ckp.json")
    if _os.path.exists(meta):
        with open(meta) as fp:
            step = _json.loads(fp.read()).get("step", 0)
    print(f"Loaded model from {model_dir} at step {step}")
    return params, state, None, step


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
