"""Trainer - mirror of lagrangebench/train/trainer.py.

Same loop as the reference (trainer.py:209-421): shuffled windows of the training split -> push-forward
unroll length (strats.py:86-109) -> ``case.preprocess`` (random-walk noise, neighbor list, features,
normalised acceleration target) -> optional no-grad unroll steps -> MSE over the non-kinematic particles
(:35-60) -> gradients summed over the batch, loss averaged (:63-89) -> AdamW(weight_decay 1e-8) on an
exponentially decaying learning rate (:183-193) -> every ``eval_steps``: validation rollouts through
``eval_rollout`` (the fused HIP loop) + checkpoint in the reference's on-disk format (:385-407).

What runs where: neighbor list, feature assembly, integrator, every evaluation rollout AND the loss step are the
HIP engine: ``lb_gns_train_loss_grad`` (csrc/lb_train.hip: forward with saved activations, masked MSE, hand-written
backward kernels, MFMA products of our own for the dense contractions) accumulates the gradients of the whole batch,
``lb_adamw_step`` applies optax.adamw on the device; weights, gradients and both moments stay in HBM as fp32.
Arithmetic of the products (include/lbhip.h, ``lb_gns_train_math_fallbacks``): by default three fp16 MFMA passes over hi / lo
splits of the fp32 operands under exact power-of-two scaling, fp32 accumulate (error per term <= 2^-22 of the operand
block's scale; gradients within 1e-4 per leaf of float64 autograd), with a range guard on the weight-gradient kernel's
activation operand that repeats a step on the exact-fp32 MFMA kernels; ``LB_TRAIN_MATH=f32`` in the environment when the
handle is created selects the exact kernels throughout (1.7x slower).
torch is used for the noise / sampling random streams and as the tensor container only.  Trainable: GNS (latent <= 128,
two to eight Linears per MLP) and, since round 5, SEGNN (lmax 1, hidden <= 32x0e+32x1o: ``lb_segnn_train_loss_grad``,
csrc/lb_train_segnn.h - the loop below is the reference's model-agnostic one); wandb logging is not wired (stdout).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from ..defaults import defaults, merge
from ..evaluate import MetricsComputer, averaged_metrics, eval_rollout
from ..evaluate.rollout import _Loader
from ..models.gns import GNS
from ..utils import (broadcast_from_batch, get_kinematic_mask, gns_params_from_haiku, gns_params_to_haiku,
                     load_haiku, save_haiku, segnn_params_from_haiku, segnn_params_to_haiku)
from .strats import push_forward_build, push_forward_sample_steps


def exponential_decay(step: int, init_value: float, transition_steps: float, decay_rate: float,
                      end_value: float) -> float:
    """optax.exponential_decay(init, transition_steps, decay_rate, end_value) as used at trainer.py:183-188:
    init * rate ** (step / transition_steps), clipped at end_value."""
    v = init_value * decay_rate ** (step / transition_steps)
    return max(v, end_value) if decay_rate < 1 else min(v, end_value)


class _ShuffledLoader:
    """DataLoader(dataset, batch_size, shuffle=True, drop_last=True, collate_fn=numpy_collate), in process."""

    def __init__(self, dataset, batch_size: int, generator: torch.Generator):
        self.dataset, self.batch_size, self.generator = dataset, batch_size, generator

    def __iter__(self):
        n = len(self.dataset)
        perm = torch.randperm(n, generator=self.generator).tolist()
        for s in range(0, n - self.batch_size + 1, self.batch_size):
            items = [self.dataset[k] for k in perm[s:s + self.batch_size]]
            yield (np.stack([it[0] for it in items]), np.stack([it[1] for it in items]))


class Trainer:
    def __init__(self, model: GNS, case, data_train, data_valid, cfg_train=None, cfg_eval=None, cfg_logging=None,
                 input_seq_length: int = defaults.model.input_seq_length, seed: int = defaults.seed):
        if isinstance(model, GNS) and (not 4 <= model._latent_size <= 128 or not 2 <= model._blocks_per_step <= 8):
            # fail HERE, before datasets and neighbor lists are set up (csrc/lb_train.hip: the training step runs on
            # 128-wide rows - narrower latents are zero-padded - with two to eight Linears per MLP)
            raise NotImplementedError(
                f"training is built for GNS with latent_size <= 128 and 2 <= num_mlp_layers <= 8 (got latent_size "
                f"{model._latent_size}, num_mlp_layers {model._blocks_per_step}); inference runs every size")
        if getattr(model, "generic", False):
            raise NotImplementedError("training is built for SEGNN in the shipped configuration (scalar_units 64, lmax_hidden = "
                                      "lmax_attributes = 1, norm None); the other switches are inference-only "
                                      "(csrc/lb_segnn_gen.hip)")
        if not hasattr(model, "train_handle"):
            raise NotImplementedError("Trainer: the model has no device training step (GNS: csrc/lb_train.hip, SEGNN: "
                                      "csrc/lb_train_segnn.h)")
        self._is_gns = isinstance(model, GNS)
        self.model, self.case, self.input_seq_length = model, case, input_seq_length
        self.cfg_train = merge(defaults.train, cfg_train)
        self.cfg_eval = merge(defaults.eval, cfg_eval)
        self.cfg_logging = merge(defaults.logging, cfg_logging)
        available = data_valid.subseq_length - input_seq_length
        assert self.cfg_eval.n_rollout_steps <= available, (
            "The loss cannot be evaluated on longer than a ground truth trajectory "
            f"({self.cfg_eval.n_rollout_steps} > {available})")
        assert self.cfg_eval.train.n_trajs <= data_valid.num_samples, (
            f"Number of requested validation trajectories exceeds the available ones "
            f"({self.cfg_eval.train.n_trajs} > {data_valid.num_samples})")
        if self.cfg_eval.train.n_trajs == -1:
            self.cfg_eval.train.n_trajs = data_valid.num_samples
        self.loss_weight = dict(self.cfg_train.loss_weight)
        self.base_key = torch.Generator()
        self.base_key.manual_seed(int(seed))
        self.loader_train = _ShuffledLoader(data_train, self.cfg_train.batch_size, self.base_key)
        self.loader_valid = _Loader(data_valid, self.cfg_eval.infer.batch_size)
        self.loader_valid.dataset = data_valid
        self.metrics_computer = MetricsComputer(self.cfg_eval.train.metrics, dist_fn=case.displacement,
                                                metadata=data_train.metadata, input_seq_length=input_seq_length,
                                                stride=self.cfg_eval.train.metrics_stride, case=case)

    def _lr(self, step: int) -> float:
        o = self.cfg_train.optimizer
        return exponential_decay(step, o.lr_start, o.lr_decay_steps, o.lr_decay_rate, o.lr_final)

    def train(self, step_max: int = defaults.train.step_max, params=None, state=None, opt_state=None,
              store_ckp: Optional[str] = None, load_ckp: Optional[str] = None, wandb_config=None
              ) -> Tuple[Dict, Dict, Dict]:
        """trainer.py:209-421.  Returns (params as numpy, state, opt_state = {"m", "v", "step"} flat AdamW moments)."""
        model, case, cfg_train, cfg_eval, cfg_logging = self.model, self.case, self.cfg_train, self.cfg_eval, self.cfg_logging
        noise_std, pushforward = cfg_train.noise_std, cfg_train.pushforward
        isl = self.input_seq_length
        key = self.base_key
        raw_batch = next(iter(self.loader_train))
        raw_sample = (raw_batch[0][0], raw_batch[1][0])
        key, features, _, neighbors = case.allocate(key, raw_sample)
        device = case.engine(1).device

        step = 0
        if params is not None:
            state = {} if state is None else state
        elif load_ckp:
            params, state, opt_state, step = load_haiku(load_ckp)
            if self._is_gns and "enc_node/linear_0" not in params:
                params = gns_params_from_haiku(params, model._mp_steps, model._blocks_per_step)
            elif not self._is_gns and "embedding_nodes" not in params:
                params = segnn_params_from_haiku(params, model)
        else:
            params, state = model.init(torch.randint(0, 2**31 - 1, (1,), generator=key).numpy(), (features, raw_sample[1]))
        B = self.loader_train.batch_size
        th = model.train_handle(case.engine(B), params)   # weights, gradients, AdamW moments: device resident
        if isinstance(opt_state, dict) and "m" in opt_state and "v" in opt_state:
            th.write("m", np.asarray(opt_state["m"], np.float32))
            # `count` = AdamW steps taken (optax's count); checkpoints of round 3 only carried the loop index `step`
            th.write("v", np.asarray(opt_state["v"], np.float32), step=int(opt_state.get("count", opt_state.get("step", step))))
        o = cfg_train.optimizer
        lw = float(self.loss_weight.get("acc", 1.0))

        def current_params():
            return model.unflatten(th.read("weights"), params)

        def opt_state_dict():
            # count: the device's AdamW step counter - NOT the loop index (it is step + 1 after an update, and differs
            # again after neighbor-list overflow `continue`s); a resumed run restores it, as optax does from opt_state
            return {"kind": "lagrangebench_amd adamw (flat blobs in the model's flatten order)", "m": th.read("m"),
                    "v": th.read("v"), "step": int(step), "count": th.step_count()}
        if store_ckp is not None:
            os.makedirs(os.path.join(store_ckp, "best"), exist_ok=True)

        push_forward = push_forward_build(model.apply, case)
        log = []
        while step < step_max + 1:
            for raw_batch in self.loader_train:
                key, unroll_steps = push_forward_sample_steps(key, step, pushforward)
                sample = (raw_batch[0], raw_batch[1])
                key, features_batch, target_batch, neighbors = case.preprocess(key, sample, noise_std, neighbors,
                                                                               unroll_steps)
                if unroll_steps > 0 and not bool(neighbors.did_buffer_overflow.sum() > 0):
                    params_np = current_params()
                    # the noisy positions the features were computed from ARE the engine's window
                    cur = case.engine(B).read_window()
                    tshift = unroll_steps
                    for _ in range(unroll_steps):
                        if neighbors.did_buffer_overflow.sum() > 0:
                            break
                        cur, neighbors, features_batch = push_forward(features_batch, cur, torch.as_tensor(raw_batch[1]),
                                                                      neighbors, params_np, state)
                    del tshift
                if neighbors.did_buffer_overflow.sum() > 0:
                    print(f"Reallocate neighbors list at step {step}")
                    ind = int(torch.argmax(neighbors.did_buffer_overflow.int()))
                    old = neighbors.max_occupancy
                    _, _, _, neighbors = case.allocate(key, (raw_batch[0][ind], raw_batch[1][ind]), noise_std)
                    print(f"From (2, {old}) to (2, {neighbors.max_occupancy})")
                    continue
                # value_and_grad of _mse vmapped over the batch, gradients summed, loss averaged (trainer.py:63-89) +
                # optax.adamw(lr(step), weight_decay 1e-8): on the engine's current window / neighbor list
                th.zero_grad()
                loss = th.loss_grad(target_batch["acc"], lw)
                th.adamw_step(self._lr(step), 0.9, 0.999, 1e-8, float(getattr(o, "weight_decay", 1e-8)))

                if step % cfg_logging.log_steps == 0:
                    step_str = str(step).zfill(len(str(int(step_max))))
                    print(f"{step_str}, train/loss: {float(loss):.5f}.")
                    log.append((step, float(loss)))
                if step % cfg_logging.eval_steps == 0 and step > 0:
                    params_np = current_params()
                    eval_metrics = eval_rollout(model_apply=model.apply, case=case, params=params_np, state=state,
                                                loader_eval=self.loader_valid, neighbors=broadcast_from_batch(neighbors, 0),
                                                metrics_computer=self.metrics_computer,
                                                n_rollout_steps=cfg_eval.n_rollout_steps, n_trajs=cfg_eval.train.n_trajs,
                                                rollout_dir=cfg_eval.rollout_dir, out_type=cfg_eval.train.out_type)
                    metrics = averaged_metrics(eval_metrics)
                    if store_ckp is not None:
                        hk = (gns_params_to_haiku(params_np, model._mp_steps, model._blocks_per_step) if self._is_gns
                              else segnn_params_to_haiku(params_np, model))
                        save_haiku(store_ckp, hk, state, opt_state_dict(), {"step": step, "loss": metrics.get("val/loss", None)})
                    print(metrics)
                    # the validation rollouts re-sized / re-used the engine: the training list is rebuilt
                    key, _, _, neighbors = case.allocate(key, raw_sample)
                step += 1
                if step == step_max + 1:
                    break
        self.loss_log = log
        out = (current_params(), state, opt_state_dict())
        th.close()
        return out
