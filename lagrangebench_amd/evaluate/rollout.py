"""Evaluation / inference rollouts - mirror of lagrangebench/evaluate/rollout.py.

Same entry points and signatures as the reference: ``_forward_eval`` (:31-75),
``_eval_batched_rollout`` (:78-178), ``eval_rollout`` (:181-308), ``infer`` (:311-399).

Two execution paths produce identical results:

* fused  - ``model_apply`` is ``GNS.apply`` of this package: the whole step loop
  (neighbor update -> features -> GNS -> integrate -> store) runs inside ``lb_rollout`` on the
  device; the host only reads the overflow flag once per launch batch.
* generic - any Python ``model_apply(params, state, (features, particle_type))`` (e.g. the
  reference's test "CheatingModel"): the loop of rollout.py:125-169 is driven from Python with
  the engine primitives, including the per-step overflow poll + re-allocation.
"""
from __future__ import annotations

import os
import pickle
import time
from functools import partial
from typing import Callable, Dict, Iterable, Optional, Tuple

import numpy as np
import torch

from ..case_setup.features import FeatureDict, NeighborList
from .._lib import LB_FORCE_BUFFER
from ..defaults import defaults, merge
from ..models.gns import GNS
from ..models.segnn import SEGNN
from ..utils import broadcast_from_batch, broadcast_to_batch, get_kinematic_mask
from .metrics import MetricsComputer, MetricsDict
from .utils import write_vtk


def _forward_eval(params, state, sample, current_positions, target_positions, model_apply: Callable,
                  case_integrate: Callable):
    """One update of the position window with a generic model - rollout.py:31-75."""
    _, particle_type = sample
    pred, state = model_apply(params, state, sample)
    next_position = case_integrate(pred, current_positions)
    kinematic_mask = get_kinematic_mask(torch.as_tensor(particle_type, device=next_position.device))
    next_position = torch.where(kinematic_mask[..., None],
                                torch.as_tensor(target_positions, device=next_position.device).to(next_position.dtype),
                                next_position)
    current_positions = torch.as_tensor(current_positions, device=next_position.device).to(next_position.dtype)
    current_positions = torch.cat([current_positions[..., 1:, :], next_position[..., None, :]], dim=-2)
    return current_positions, state


def _gns_of(model_apply) -> Optional[GNS]:
    fn = model_apply
    while isinstance(fn, partial):
        fn = fn.func
    owner = getattr(fn, "__self__", None)
    return owner if isinstance(owner, (GNS, SEGNN)) else None


def _eval_batched_rollout(forward_eval_vmap: Callable, preprocess_eval_vmap: Callable, case, params, state,
                          traj_batch_i: Tuple, neighbors: NeighborList, metrics_computer_vmap: Callable,
                          n_rollout_steps: int, t_window: int, n_extrap_steps: int = 0):
    """Rollout of one batch of trajectories - rollout.py:78-178.

    ``forward_eval_vmap`` / ``preprocess_eval_vmap`` are natively batched here (no vmap needed).
    If ``forward_eval_vmap`` was built by :func:`eval_rollout` around ``GNS.apply`` the fused
    device loop is used; otherwise the generic loop below calls the two callables per step
    exactly like the reference does."""
    pos_input_batch, particle_type_batch = traj_batch_i
    pos_input_batch = torch.as_tensor(np.asarray(pos_input_batch)) if not isinstance(pos_input_batch, torch.Tensor) else pos_input_batch
    particle_type_batch = torch.as_tensor(np.asarray(particle_type_batch)) if not isinstance(particle_type_batch, torch.Tensor) else particle_type_batch
    B, n_nodes_max, _, dim = pos_input_batch.shape
    if n_rollout_steps == -1:
        n_rollout_steps = pos_input_batch.shape[2] - t_window
    traj_len = n_rollout_steps + n_extrap_steps
    eng = case.engine(B)
    traj = eng.prepare_traj(pos_input_batch)
    ptype = particle_type_batch.to(eng.device)
    target_positions_batch = traj[:, :, t_window:t_window + traj_len]

    gns = getattr(forward_eval_vmap, "_lb_gns", None)
    if gns is not None and eng.force is not None and eng.force.kind == LB_FORCE_BUFFER:
        # a host-evaluated external_force_fn (features.py:105-107 with an arbitrary callable) has to
        # be refreshed between steps: lb_rollout cannot call back into Python, drive the loop here
        gns = None
    if gns is not None:
        # ---- fused path: lb_rollout ---------------------------------------------------
        eng.set_particle_type(ptype)
        if neighbors is not None and (eng.e_cap, eng.cell_capacity) != (neighbors.max_occupancy, neighbors.cell_capacity):
            eng.nl_set_capacity(neighbors.cell_capacity, neighbors.max_occupancy)
        predictions_batch, n_realloc = eng.rollout(gns.handle(eng, params), traj, traj_len)
        if n_realloc:
            print(f"(eval) Reallocated the neighbors list {n_realloc}x; capacity now (2, {eng.e_cap})")
        eng.load_window(traj, t0=0, step=0)  # leave a defined state behind for the returned list
        eng.nl_update()
        neighbors_out = NeighborList(eng, False)
    else:
        # ---- generic path: the reference's Python loop (rollout.py:125-169) -------------
        current_positions_batch = traj[:, :, 0:t_window]
        predictions_batch = torch.zeros((B, traj_len, n_nodes_max, dim), dtype=torch.float64, device=eng.device)
        neighbors_batch = broadcast_to_batch(neighbors, B)
        step = 0
        while step < traj_len:
            sample_batch = (current_positions_batch, ptype)
            features_batch, neighbors_batch = preprocess_eval_vmap(sample_batch, neighbors_batch)
            if neighbors_batch.did_buffer_overflow.sum() > 0:  # host sync, as in the reference
                print(f"(eval) Reallocate neighbors list at step {step}")
                _, nbrs_temp = case.allocate_eval(sample_batch)
                print(f"(eval) From (2, {neighbors_batch.max_occupancy}) to (2, {nbrs_temp.max_occupancy})")
                neighbors_batch = nbrs_temp
                continue
            tstep = min(step, target_positions_batch.shape[2] - 1)  # JAX clamps the OOB gather
            current_positions_batch, state = forward_eval_vmap(
                params, state, (features_batch, ptype), current_positions_batch,
                target_positions_batch[:, :, tstep])
            predictions_batch[:, step] = current_positions_batch[:, :, -1]
            step += 1
        neighbors_out = broadcast_from_batch(neighbors_batch, 0)

    # (batch, n_nodes, time, dim) -> (batch, time, n_nodes, dim); metrics on the non-extrapolated part
    target_t = target_positions_batch.permute(0, 2, 1, 3).contiguous()
    metrics_batch = metrics_computer_vmap(predictions_batch[:, :n_rollout_steps], target_t)
    return predictions_batch, metrics_batch, neighbors_out


def eval_rollout(model_apply: Callable, case, params, state, loader_eval: Iterable, neighbors: NeighborList,
                 metrics_computer: MetricsComputer, n_rollout_steps: int, n_trajs: int, rollout_dir: str,
                 out_type: str = "none", n_extrap_steps: int = 0) -> MetricsDict:
    """Compute rollouts + metrics for ``n_trajs`` trajectories - rollout.py:181-308."""
    batch_size = loader_eval.batch_size
    t_window = loader_eval.dataset.input_seq_length
    eval_metrics = {}
    if rollout_dir is not None:
        os.makedirs(rollout_dir, exist_ok=True)

    forward_eval = partial(_forward_eval, model_apply=model_apply, case_integrate=case.integrate)
    gns = _gns_of(model_apply)
    if gns is not None:
        forward_eval._lb_gns = gns  # type: ignore[attr-defined]  -> fused device loop
    preprocess_eval = case.preprocess_eval
    if getattr(metrics_computer, "_case", None) is None:
        metrics_computer._case = case

    # One process per GPU (torchrun): batch b of the loader belongs to rank b % world - trajectories
    # never interact (rollout.py:226-230), so there is no data-path collective; the per-rollout
    # metric dictionaries are all-gathered at the end (lagrangebench_amd/dist.py).
    rank, world = 0, 1
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
    if hasattr(loader_eval, "set_shard"):
        loader_eval.set_shard(rank, world)  # batches of other ranks are not even read from disk
    for i, traj_batch_i in enumerate(loader_eval):
        n_traj_left = n_trajs - i * batch_size
        if n_traj_left <= 0:
            break
        if i % world != rank or traj_batch_i is None:
            continue
        if n_traj_left < batch_size:
            traj_batch_i = tuple(x[:n_traj_left] for x in traj_batch_i)
        example_rollout_batch, metrics_batch, neighbors = _eval_batched_rollout(
            forward_eval_vmap=forward_eval, preprocess_eval_vmap=preprocess_eval, case=case,
            params=params, state=state, traj_batch_i=traj_batch_i, neighbors=neighbors,
            metrics_computer_vmap=metrics_computer, n_rollout_steps=n_rollout_steps,
            t_window=t_window, n_extrap_steps=n_extrap_steps)
        current_batch_size = len(traj_batch_i[0])
        for j in range(current_batch_size):
            ind = i * batch_size + j
            eval_metrics[f"rollout_{ind}"] = broadcast_from_batch(metrics_batch, j)
        if rollout_dir is not None and out_type in ("pkl", "vtk"):
            pos_np = np.asarray(traj_batch_i[0])
            for j in range(current_batch_size):
                pos_input = np.transpose(pos_np[j], (1, 0, 2))
                example_full = np.concatenate([pos_input[:t_window], example_rollout_batch[j].cpu().numpy()])
                example = {"predicted_rollout": example_full, "ground_truth_rollout": pos_input,
                           "particle_type": np.asarray(traj_batch_i[1][j])}
                file_prefix = os.path.join(rollout_dir, f"rollout_{i * batch_size + j}")
                if out_type == "vtk":  # one file per time step, rollout.py:278-292
                    for k in range(example_full.shape[0]):
                        write_vtk({"r": example_full[k], "tag": example["particle_type"]}, f"{file_prefix}_{k}.vtk")
                    for k in range(pos_input.shape[0]):
                        write_vtk({"r": pos_input[k], "tag": example["particle_type"]}, f"{file_prefix}_ref_{k}.vtk")
                else:
                    with open(f"{file_prefix}.pkl", "wb") as f:
                        pickle.dump(example, f)
    if world > 1:
        from ..dist import gather_metric_dicts
        eval_metrics = gather_metric_dicts(eval_metrics)

    if rollout_dir is not None and rank == 0:
        t = time.strftime("%Y_%m_%d_%H_%M_%S", time.localtime())
        def _cpu(x):
            return {k: _cpu(v) for k, v in x.items()} if isinstance(x, dict) else x.cpu().numpy()
        cpu = _cpu(eval_metrics)
        with open(f"{rollout_dir}/metrics{t}.pkl", "wb") as f:
            pickle.dump(cpu, f)
    return eval_metrics


class _Loader:
    """Minimal stand-in for torch DataLoader(dataset, batch_size, collate_fn=numpy_collate)."""

    def __init__(self, dataset, batch_size: int):
        self.dataset, self.batch_size = dataset, batch_size
        self._rank, self._world = 0, 1

    def set_shard(self, rank: int, world: int) -> None:
        """Batches owned by other ranks are yielded as None (keeps the batch numbering)."""
        self._rank, self._world = rank, world

    def __iter__(self):
        n = len(self.dataset)
        for s in range(0, n, self.batch_size):
            if (s // self.batch_size) % self._world != self._rank:
                yield None
                continue
            items = [self.dataset[k] for k in range(s, min(n, s + self.batch_size))]
            yield (np.stack([it[0] for it in items]), np.stack([it[1] for it in items]))


def infer(model, case, data_test, params=None, state=None, load_ckp: Optional[str] = None,
          cfg_eval_infer=None, rollout_dir: Optional[str] = None,
          n_rollout_steps: int = defaults.eval.n_rollout_steps, seed: int = defaults.seed):
    """Infer on a dataset and compute metrics - rollout.py:311-399."""
    assert params is not None or load_ckp is not None, \
        "Either params or a load_ckp directory must be provided for inference."
    cfg = merge(defaults.eval.infer, cfg_eval_infer)
    n_trajs = cfg.n_trajs
    if n_trajs == -1:
        n_trajs = data_test.num_samples
    if params is None:
        # rollout.py:359 load_haiku(load_ckp); a Haiku-named GNS tree is mapped onto the engine's layout
        from ..utils import gns_params_from_haiku, load_haiku
        params, state, _, _ = load_haiku(load_ckp)
        if isinstance(model, GNS) and "enc_node/linear_0" not in params:
            params = gns_params_from_haiku(params, model._mp_steps, model._blocks_per_step)
        if isinstance(model, SEGNN) and "embedding_nodes" not in params:
            from ..utils import segnn_params_from_haiku
            params = segnn_params_from_haiku(params, model)
    if state is None:
        state = {}
    loader_test = _Loader(data_test, cfg.batch_size)
    metrics_computer = MetricsComputer(cfg.metrics, dist_fn=case.displacement, metadata=data_test.metadata,
                                       input_seq_length=data_test.input_seq_length,
                                       stride=cfg.metrics_stride, case=case)
    pos0, ptype0 = data_test[0]
    _, _, _, neighbors = case.allocate(None, (pos0, ptype0))
    return eval_rollout(model_apply=model.apply, case=case, metrics_computer=metrics_computer, params=params,
                        state=state, neighbors=neighbors, loader_eval=loader_test,
                        n_rollout_steps=n_rollout_steps, n_trajs=n_trajs, rollout_dir=rollout_dir,
                        out_type=cfg.out_type, n_extrap_steps=cfg.n_extrap_steps)
