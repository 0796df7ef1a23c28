"""Runner: mirror of lagrangebench/runner.py for the inference route (`mode: infer`).

``train_or_infer(cfg)`` (runner.py:25-143), ``setup_data`` (:146-189) and ``setup_model`` (:192-292)
keep their signatures.  ``cfg`` is a nested mapping with the reference's keys (a dict or anything
dict-like such as an OmegaConf DictConfig); missing keys fall back to ``defaults``.  `mode: train | all`
runs ``train.Trainer`` (GNS and SEGNN).
"""
from __future__ import annotations

import os
import os.path as osp
from typing import Callable, Dict, Optional, Tuple, Type

import numpy as np

from . import models
from .case_setup import case_builder
from .data import H5Dataset
from .defaults import defaults, merge
from .evaluate import averaged_metrics, infer
from .utils import NodeType

_RUN_DEFAULTS = {
    "mode": "infer", "load_ckp": None, "dtype": "float64", "seed": 0,
    "dataset": {"src": None, "name": None},
    "model": dict(defaults.model),
    "train": dict(defaults.train),
    "eval": {"test": False, "n_rollout_steps": 20, "rollout_dir": None,
             "infer": dict(defaults.eval.infer), "train": dict(defaults.eval.train)},
    "logging": dict(defaults.logging),
    "neighbors": dict(defaults.neighbors),
}


def setup_data(cfg) -> Tuple[H5Dataset, H5Dataset, H5Dataset]:
    """runner.py:146-189."""
    cfg = merge(_RUN_DEFAULTS, cfg)
    dataset_path = cfg.dataset.src
    if not osp.isabs(dataset_path):
        dataset_path = osp.join(os.getcwd(), dataset_path)
    if cfg.logging.ckp_dir is not None and cfg.mode in ("train", "all"):  # (inference writes no checkpoint)
        os.makedirs(cfg.logging.ckp_dir, exist_ok=True)
    if cfg.eval.rollout_dir is not None:
        os.makedirs(cfg.eval.rollout_dir, exist_ok=True)
    kw = dict(dataset_path=dataset_path, name=cfg.dataset.name, input_seq_length=cfg.model.input_seq_length,
              nl_backend=cfg.neighbors.backend)
    data_train = H5Dataset("train", extra_seq_length=cfg.train.pushforward.unrolls[-1], **kw)
    data_valid = H5Dataset("valid", extra_seq_length=cfg.eval.n_rollout_steps, **kw)
    data_test = H5Dataset("test", extra_seq_length=cfg.eval.n_rollout_steps, **kw)
    return data_train, data_valid, data_test


def setup_model(cfg, metadata: Dict, homogeneous_particles: bool = False, has_external_force: bool = False,
                normalization_stats: Optional[Dict] = None) -> Tuple[Callable, Type]:
    """runner.py:192-292.  Returns (model, MODEL class); the model already exposes .init/.apply
    (no hk.transform_with_state step)."""
    cfg = merge(_RUN_DEFAULTS, cfg)
    name = str(cfg.model.name).lower()
    if name == "gns":
        model = models.GNS(
            particle_dimension=metadata["dim"], latent_size=cfg.model.latent_dim,
            blocks_per_step=cfg.model.num_mlp_layers, num_mp_steps=cfg.model.num_mp_steps,
            num_particle_types=NodeType.SIZE, particle_type_embedding_size=16)
        return model, models.GNS
    if name == "segnn":
        # runner.py:217-245: Hx1o vel, [2x1o boundary], [1x1o force], [Hx0e |vel|], [9x0e type]
        isl = cfg.model.input_seq_length
        irreps = models.node_irreps(metadata, isl, has_external_force, cfg.model.magnitude_features,
                                    homogeneous_particles)
        model = models.SEGNN(
            node_features_irreps=irreps, edge_features_irreps="1x1o+1x0e",
            scalar_units=cfg.model.latent_dim, lmax_hidden=cfg.model.lmax_hidden,
            lmax_attributes=cfg.model.lmax_attributes, output_irreps="1x1o",
            num_mp_steps=cfg.model.num_mp_steps, n_vels=isl - 1,
            velocity_aggregate=cfg.model.velocity_aggregate, homogeneous_particles=homogeneous_particles,
            blocks_per_step=cfg.model.num_mlp_layers, norm=cfg.model.segnn_norm)
        return model, models.SEGNN
    raise NotImplementedError(f"model {cfg.model.name!r}: 'gns' and 'segnn' are built (egnn/painn/linear are not)")


def train_or_infer(cfg):
    """runner.py:25-143, inference route."""
    cfg = merge(_RUN_DEFAULTS, cfg)
    mode = cfg.mode
    if mode not in ("train", "infer", "all"):
        raise ValueError("mode must be one of 'train', 'infer', 'all'")
    if cfg.dtype not in ("float64", "float32"):
        raise NotImplementedError("dtype must be float64 (the reference default) or float32")
    data_train, data_valid, data_test = setup_data(cfg)
    metadata = data_train.metadata
    bounds = np.array(metadata["bounds"])
    box = bounds[:, 1] - bounds[:, 0]
    case = case_builder(box=box, metadata=metadata, input_seq_length=cfg.model.input_seq_length,
                        cfg_neighbors=cfg.neighbors, cfg_model=cfg.model, noise_std=cfg.train.noise_std,
                        external_force_fn=data_train.external_force_fn, dtype=cfg.dtype)
    _, particle_type = data_train[0]
    model, _ = setup_model(cfg, metadata=metadata,
                           homogeneous_particles=particle_type.max() == particle_type.min(),
                           has_external_force=data_train.external_force_fn is not None,
                           normalization_stats=case.normalization_stats)
    if mode in ("train", "all"):  # runner.py:74-117
        import time
        from .train import Trainer
        print("Start training...")
        if cfg.logging.run_name is None:
            cfg.logging.run_name = f"{cfg.model.name}_{data_train.name}_{time.strftime('%Y%m%d-%H%M%S')}"
        store_ckp = os.path.join(cfg.logging.ckp_dir, cfg.logging.run_name) if cfg.logging.ckp_dir else None
        trainer = Trainer(model, case, data_train, data_valid, cfg.train, cfg.eval, cfg.logging,
                          input_seq_length=cfg.model.input_seq_length, seed=cfg.seed)
        trainer.train(step_max=cfg.train.step_max, load_ckp=cfg.load_ckp, store_ckp=store_ckp)
        if mode == "train":
            return 0
        cfg.load_ckp = os.path.join(store_ckp, "best") if store_ckp else cfg.load_ckp
    print("Start inference...")
    model_dir = cfg.load_ckp
    assert model_dir, "model_dir must be specified for inference."
    is_test = cfg.eval.test
    metrics = infer(model, case, data_test if is_test else data_valid, load_ckp=model_dir,
                    cfg_eval_infer=cfg.eval.infer, rollout_dir=cfg.eval.rollout_dir,
                    n_rollout_steps=cfg.eval.n_rollout_steps, seed=cfg.seed)
    split = "test" if is_test else "valid"
    print(f"Metrics of {model_dir} on {split} split:")
    print(averaged_metrics(metrics))
    return 0
