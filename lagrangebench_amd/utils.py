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
