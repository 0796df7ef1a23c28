"""Trainer - mirror of lagrangebench/train/trainer.py.

Same loop as the reference (trainer.py:209-421): shuffled windows of the training split -> push-forward
unroll length (strats.py:86-109) -> ``case.preprocess`` (random-walk noise, neighbor list, features,
normalised acceleration target) -> optional no-grad unroll steps -> MSE over the non-kinematic particles
(:35-60) -> gradients summed over the batch, loss averaged (:63-89) -> AdamW(weight_decay 1e-8) on an
exponentially decaying learning rate (:183-193) -> every ``eval_steps``: validation rollouts through
``eval_rollout`` (the fused HIP loop) + checkpoint in the reference's on-disk format (:385-407).

What runs where: neighbor list, feature assembly, integrator and every evaluation rollout are the HIP
engine; the loss step differentiates ``models/gns_torch.py`` (a torch restatement of the GNS forward, fp32
on the same GPU) with torch.autograd and steps ``torch.optim.AdamW`` - there are no hand-written backward
kernels yet, which is why DESIGN.md lists this row as partial.  Only GNS is trainable; wandb logging is
not wired (stdout, as the reference's default).
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
from ..models.gns_torch import gns_apply_torch, gns_inputs_from_features, params_to_numpy, params_to_torch
from ..utils import (broadcast_from_batch, get_kinematic_mask, gns_params_from_haiku, gns_params_to_haiku,
                     load_haiku, save_haiku)
from .strats import push_forward_build, push_forward_sample_steps


def exponential_decay(step: int, init_value: float, transition_steps: float, decay_rate: float,
                      end_value: float) -> float:
    """optax.exponential_decay(init, transition_steps, decay_rate, end_value) as used at trainer.py:183-188:
    init * rate ** (step / transition_steps), clipped at end_value."""
    v = init_value * decay_rate ** (step / transition_steps)
    return max(v, end_value) if decay_rate < 1 else min(v, end_value)


def _mse(params_t, features, particle_type, target, model: GNS, loss_weight: Dict[str, float]):
    """trainer.py:35-60 for one trajectory of the batch: weighted squared error of every predicted
    quantity, summed over dim, averaged over the non-kinematic particles."""
    node, edge, snd, rcv, pt = gns_inputs_from_features(features, particle_type)
    pred = {"acc": gns_apply_torch(params_t, node, edge, snd, rcv, pt, model._mp_steps, model._blocks_per_step)}
    non_kin = ~get_kinematic_mask(pt)
    total = sum(float(loss_weight[k]) * ((pred[k] - target[k].to(pred[k].dtype)) ** 2).sum(dim=-1) for k in pred)
    total = torch.where(non_kin, total, torch.zeros_like(total))
    return total.sum() / non_kin.sum()


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
        if not isinstance(model, GNS):
            raise NotImplementedError("Trainer: only GNS has a differentiable forward (models/gns_torch.py)")
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
        """trainer.py:209-421.  Returns (params as numpy, state, opt_state = torch AdamW state_dict)."""
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
            if "enc_node/linear_0" not in params:
                params = gns_params_from_haiku(params, model._mp_steps, model._blocks_per_step)
        else:
            params, state = model.init(torch.randint(0, 2**31 - 1, (1,), generator=key).numpy(), (features, raw_sample[1]))
        params_t = params_to_torch(params, device=device, requires_grad=True)
        leaves = [v for mod in sorted(params_t) for _, v in sorted(params_t[mod].items())]
        opt = torch.optim.AdamW(leaves, lr=self._lr(step), betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-8)
        if isinstance(opt_state, dict) and "state" in opt_state:
            # (a checkpoint stores the moments as numpy arrays: utils.save_haiku)
            def _t(x):
                if isinstance(x, np.ndarray):
                    return torch.as_tensor(x)
                if isinstance(x, dict):
                    return {k: _t(v) for k, v in x.items()}
                if isinstance(x, list):
                    return [_t(v) for v in x]
                return x
            opt.load_state_dict(_t(opt_state))
        if store_ckp is not None:
            os.makedirs(os.path.join(store_ckp, "best"), exist_ok=True)

        push_forward = push_forward_build(model.apply, case)
        B = self.loader_train.batch_size
        log = []
        while step < step_max + 1:
            for raw_batch in self.loader_train:
                key, unroll_steps = push_forward_sample_steps(key, step, pushforward)
                sample = (raw_batch[0], raw_batch[1])
                key, features_batch, target_batch, neighbors = case.preprocess(key, sample, noise_std, neighbors,
                                                                               unroll_steps)
                if unroll_steps > 0 and not bool(neighbors.did_buffer_overflow.sum() > 0):
                    params_np = params_to_numpy(params_t)
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
                features_batch.materialize()
                for g in opt.param_groups:
                    g["lr"] = self._lr(step)
                opt.zero_grad(set_to_none=True)
                losses = []
                for b in range(B):
                    fb = {k: features_batch[k][b] for k in features_batch.keys()}
                    tb = {"acc": target_batch["acc"][b]}
                    lb = _mse(params_t, fb, torch.as_tensor(raw_batch[1][b]), tb, model, self.loss_weight)
                    lb.backward()                      # gradients summed over the batch (trainer.py:82)
                    losses.append(lb.detach())
                opt.step()
                loss = torch.stack(losses).mean()      # loss averaged over the batch (trainer.py:84)

                if step % cfg_logging.log_steps == 0:
                    step_str = str(step).zfill(len(str(int(step_max))))
                    print(f"{step_str}, train/loss: {float(loss):.5f}.")
                    log.append((step, float(loss)))
                if step % cfg_logging.eval_steps == 0 and step > 0:
                    params_np = params_to_numpy(params_t)
                    eval_metrics = eval_rollout(model_apply=model.apply, case=case, params=params_np, state=state,
                                                loader_eval=self.loader_valid, neighbors=broadcast_from_batch(neighbors, 0),
                                                metrics_computer=self.metrics_computer,
                                                n_rollout_steps=cfg_eval.n_rollout_steps, n_trajs=cfg_eval.train.n_trajs,
                                                rollout_dir=cfg_eval.rollout_dir, out_type=cfg_eval.train.out_type)
                    metrics = averaged_metrics(eval_metrics)
                    if store_ckp is not None:
                        save_haiku(store_ckp, gns_params_to_haiku(params_np, model._mp_steps, model._blocks_per_step),
                                   state, opt.state_dict(), {"step": step, "loss": metrics.get("val/loss", None)})
                    print(metrics)
                    # the validation rollouts re-sized / re-used the engine: the training list is rebuilt
                    key, _, _, neighbors = case.allocate(key, raw_sample)
                step += 1
                if step == step_max + 1:
                    break
        self.loss_log = log
        return params_to_numpy(params_t), state, opt.state_dict()
