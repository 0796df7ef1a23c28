"""Training tricks - mirror of lagrangebench/train/strats.py: random-walk noise (:12-83) and the
push-forward schedule / unroll (:86-161).

Random numbers: the reference threads a jax.random key through these functions; here the "key" is a
``torch.Generator`` (CPU) that is advanced in place and returned, so call sites keep the reference's
``key, x = fn(key, ...)`` shape.  (Bit-identical noise to JAX's threefry stream is not a goal: the
statistics are what training depends on, and they are what the tests pin.)
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np
import torch

from ..utils import get_kinematic_mask


def _as_gen(key) -> torch.Generator:
    if isinstance(key, torch.Generator):
        return key
    g = torch.Generator()
    g.manual_seed(int(np.asarray(key).ravel()[-1]) if key is not None else 0)
    return g


def _get_random_walk_noise_for_pos_sequence(key, position_sequence: torch.Tensor, noise_std_last_step: float):
    """strats.py:61-83: N(0, 1) velocity noise scaled so that the LAST step of its random walk has standard
    deviation `noise_std_last_step`, accumulated twice (velocity walk -> position walk), zero on frame 0."""
    g = _as_gen(key)
    shape = list(position_sequence.shape)
    shape[-2] -= 1                       # (..., n_velocities, dim)
    n_vel = shape[-2]
    noise = torch.randn(shape, generator=g, dtype=torch.float64)
    noise = noise * (noise_std_last_step / n_vel ** 0.5)
    vel_noise = torch.cumsum(noise, dim=-2)
    pos_noise = torch.cat([torch.zeros_like(vel_noise[..., 0:1, :]), torch.cumsum(vel_noise, dim=-2)], dim=-2)
    return g, pos_noise.to(position_sequence.device, position_sequence.dtype)


def add_gns_noise(key, pos_input: torch.Tensor, particle_type: torch.Tensor, input_seq_length: int,
                  noise_std: float, shift_fn: Callable) -> Tuple[torch.Generator, torch.Tensor]:
    """strats.py:12-58.  pos_input (N, T, dim) or batched (B, N, T, dim); the noise of the last input frame
    is carried onto every later (target) frame; kinematic particles stay clean."""
    isl = input_seq_length
    key, noise = _get_random_walk_noise_for_pos_sequence(key, pos_input[..., :isl, :], noise_std)
    kin = get_kinematic_mask(torch.as_tensor(particle_type, device=pos_input.device))
    noise = torch.where(kin[..., None, None], torch.zeros_like(noise), noise)
    n_targets = pos_input.shape[-2] - isl
    tail = noise[..., -1:, :].expand(*noise.shape[:-2], n_targets, noise.shape[-1])
    noise = torch.cat([noise, tail], dim=-2)
    return key, shift_fn(pos_input, noise)


def push_forward_sample_steps(key, step: int, pushforward) -> Tuple[torch.Generator, int]:
    """strats.py:86-109: unroll stages unlock when `step` passes pushforward.steps[i]; among the unlocked
    stages the unroll length is drawn with the relative probabilities pushforward.probs."""
    g = _as_gen(key)
    steps = list(pushforward["steps"])
    assert all(steps[i] <= steps[i + 1] for i in range(len(steps) - 1))
    idx = int(sum(step > s for s in steps))
    unrolls = list(pushforward["unrolls"])[:idx]
    probs = torch.tensor(list(pushforward["probs"])[:idx], dtype=torch.float64)
    pick = int(torch.multinomial(probs / probs.sum(), 1, generator=g))
    return g, int(unrolls[pick])


def push_forward_build(model_apply: Callable, case) -> Callable:
    """strats.py:112-161: one solver step WITHOUT gradients (model -> integrate -> window shift ->
    preprocess_eval), used to unroll the input of the loss step."""

    @torch.no_grad()
    def push_forward_fn(features, current_pos, particle_type, neighbors, params, state):
        pred, _ = model_apply(params, state, (features, particle_type))
        next_pos = case.integrate(pred, current_pos)
        cur = torch.as_tensor(current_pos, device=next_pos.device).to(next_pos.dtype)
        current_pos = torch.cat([cur[..., 1:, :], next_pos[..., None, :]], dim=-2)
        features, neighbors = case.preprocess_eval((current_pos, particle_type), neighbors)
        return current_pos, neighbors, features

    return push_forward_fn
