"""SEGNN training step (SURVEY.md section 8f row N4 for config 5's model; reference: train/trainer.py:35-89 is model-agnostic).

CPU: the differentiable torch restatement (oracle/segnn_torch.py) equals the NumPy oracle and its autograd gradients agree with
finite differences.  GPU: the device step (csrc/lb_train_segnn.h behind lb_segnn_train_*) - loss and every weight / bias gradient
against float64 autograd of that restatement on engine-built graphs, bit-reproducibility, AdamW.
Parity with e3nn-jax itself is unpinned (oracle/segnn_oracle.py header).
"""
import numpy as np
import pytest
import torch

from oracle import segnn_oracle as S
from oracle import segnn_torch as ST
from tests.test_segnn import _random_graph_features, _setup


def _loss_torch(pt_params, feats, ptype, n_vels, homog, target, blocks, layers):
    node, nattr, eattr, msg, snd, rcv, dim = ST.inputs_from_features(feats, ptype, n_vels, homog)
    pred = ST.segnn_apply_torch(pt_params, node, nattr, eattr, msg, snd, rcv, dim, blocks, layers)
    # trainer.py:35-60 _mse: weighted squared error summed over dim, masked to non-kinematic particles, / their number
    from oracle import lb_oracle as O
    mask = torch.tensor(~O.get_kinematic_mask(np.asarray(ptype)))
    se = ((pred - torch.tensor(target, dtype=pred.dtype)) ** 2).sum(dim=1)
    return (se * mask).sum() / mask.sum(), pred


def test_torch_restatement_matches_numpy_oracle():
    n, K = 40, 5
    pt = np.random.default_rng(2).integers(0, 3, n)
    p = S.segnn_init(np.random.default_rng(0), node_ns=K + 9, node_nv=K + 3, num_mp_steps=3, random_bias=True)
    feats = _random_graph_features(np.eye(3))
    with S.precision(np.float64):
        ref = S.segnn_apply(p, feats, pt, K, False)["acc"]
    node, nattr, eattr, msg, snd, rcv, dim = ST.inputs_from_features(feats, pt, K, False)
    got = ST.segnn_apply_torch(ST.params_to_torch(p), node, nattr, eattr, msg, snd, rcv, dim, 2, 3).numpy()
    assert np.abs(ref).max() > 1e-3
    assert np.abs(got - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())


def test_torch_gradients_match_finite_differences():
    n, K = 30, 3
    pt = np.zeros(n, np.int64)
    p = S.segnn_init(np.random.default_rng(3), node_ns=K, node_nv=K + 3, num_mp_steps=2, random_bias=True)
    feats = _random_graph_features(np.eye(3), n=n, K=K, E=120)
    target = np.random.default_rng(4).standard_normal((n, 3))
    tp = ST.params_to_torch(p, requires_grad=True)
    loss, _ = _loss_torch(tp, feats, pt, K, True, target, 2, 2)
    loss.backward()
    rng = np.random.default_rng(5)
    for name in ["embedding_nodes", "layer_0/message_0", "layer_1/update_1", "readout_1", "output"]:
        for leaf in ("ws", "wv", "b"):
            w = tp[name][leaf]
            if w.numel() == 0:
                continue
            idx = tuple(int(rng.integers(0, s)) for s in w.shape)
            h = 1e-6
            with torch.no_grad():
                old = w[idx].item()
                w[idx] = old + h
                lp, _ = _loss_torch(tp, feats, pt, K, True, target, 2, 2)
                w[idx] = old - h
                lm, _ = _loss_torch(tp, feats, pt, K, True, target, 2, 2)
                w[idx] = old
            fd = (lp.item() - lm.item()) / (2 * h)
            assert abs(fd - w.grad[idx].item()) < 1e-5 * max(1.0, abs(fd)), (name, leaf, fd, w.grad[idx].item())


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name,scale,L,B,blocks", [("small2d", 1.0, 2, 2, 2), ("small3d", 1.0, 3, 1, 2), ("dam2d", 0.3, 2, 2, 2),
                                                   ("rpf2d", 0.5, 10, 1, 2), ("small2d", 1.0, 2, 1, 1), ("small3d", 1.0, 2, 1, 3)])
def test_hip_segnn_gradients_match_torch_autograd(name, scale, L, B, blocks):
    """The device step (lb_segnn_train_loss_grad) against float64 autograd of oracle/segnn_torch.py on engine-built graphs:
    prediction, loss (mean over the batch) and every weight / bias gradient (summed over the batch, trainer.py:63-89) within
    1e-4 of the leaf's largest entry; two runs give the same bits; one AdamW step against torch.optim.AdamW."""
    from lagrangebench_amd.data import make_case
    from lagrangebench_amd.models import SEGNN, node_irreps
    from oracle import lb_oracle as O
    from tests._common import hip_case, oracle_case
    ds = make_case(name, n_trajs=B, extra_seq_length=3, scale=scale)
    ds.magnitude_features = True
    isl, dim = ds.input_seq_length, len(ds.box)
    homog = bool(np.all(ds[0][1] == 0))
    irr = node_irreps(ds.metadata, isl, ds.external_force_fn is not None, True, homog)
    model = SEGNN(irr, "1x1o+1x0e", 64, 1, 1, "1x1o", num_mp_steps=L, n_vels=isl - 1, homogeneous_particles=homog,
                  blocks_per_step=blocks)
    params = S.segnn_init(np.random.default_rng(11), node_ns=model._node_ns, node_nv=model._node_nv, num_mp_steps=L,
                          blocks_per_step=blocks, random_bias=True)
    params = {k: v for k, v in params.items() if isinstance(v, dict)}
    ocase, hcase = oracle_case(ds), hip_case(ds)
    pos = np.stack([ds[b][0] for b in range(B)])
    pt = np.stack([ds[b][1] for b in range(B)])
    feats, _ = hcase.allocate_eval((pos[:, :, :isl], pt))
    eng = feats.engine
    N = pos.shape[1]
    target = torch.randn((B, N, dim), generator=torch.Generator().manual_seed(5))
    th = model.train_handle(eng, params)
    th.zero_grad()
    loss_h, pred_h = th.loss_grad(target, 1.0, want_pred=True)
    g_flat = th.read("grads")
    for _ in range(2):  # no floating-point atomics anywhere in the step
        th.zero_grad()
        loss_2, _ = th.loss_grad(target, 1.0, want_pred=True)
        assert loss_2 == loss_h and np.array_equal(th.read("grads"), g_flat)
    g_h = model.unflatten(g_flat)
    assert np.array_equal(model.flatten(model.unflatten(th.read("weights"))), model.flatten(params))  # blob <-> device layout

    tp = ST.params_to_torch(params, requires_grad=True)
    losses = []
    for b in range(B):
        of, _ = ocase.allocate_eval((pos[b][:, :isl].astype(np.float64), pt[b]))
        node, nattr, eattr, msg, snd, rcv, d3 = ST.inputs_from_features(of, pt[b], isl - 1, homog)
        pred = ST.segnn_apply_torch(tp, node, nattr, eattr, msg, snd, rcv, d3, blocks, L)
        ph = pred_h[b].detach().cpu().double()
        assert float((pred.detach() - ph).abs().max() / pred.detach().abs().max()) < 1e-5
        nk = torch.tensor(~O.get_kinematic_mask(pt[b]))
        tot = ((pred - target[b].double()) ** 2).sum(dim=-1)
        lb = torch.where(nk, tot, torch.zeros_like(tot)).sum() / nk.sum()
        lb.backward()
        losses.append(float(lb))
    assert abs(loss_h - np.mean(losses)) <= 1e-5 * abs(np.mean(losses)), (loss_h, losses)
    worst = 0.0
    for blk, leaves in tp.items():
        for leaf, v in leaves.items():
            if v.numel() == 0:
                continue
            ref = v.grad.numpy()
            err = np.abs(g_h[blk][leaf] - ref).max() / max(np.abs(ref).max(), 1e-30)
            worst = max(worst, err)
            assert err < 1e-4, (blk, leaf, err)
    print(f"[segnn grad {name} L={L} B={B} blocks={blocks}] loss {loss_h:.6f}, worst relative gradient error {worst:.2e}")

    for blk, lv in tp.items():
        for leaf, v in lv.items():
            v.grad = torch.as_tensor(g_h[blk][leaf]).double()
    leaves = [v for blk in sorted(tp) for _, v in sorted(tp[blk].items()) if v.numel()]
    opt = torch.optim.AdamW(leaves, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    opt.step()
    th.adamw_step(1e-3, 0.9, 0.999, 1e-8, 1e-2)
    w_h = model.unflatten(th.read("weights"))
    for blk, lv in tp.items():
        for leaf, v in lv.items():
            if v.numel():
                ref = v.detach().numpy()
                assert np.abs(w_h[blk][leaf] - ref).max() <= 2e-6 * max(np.abs(ref).max(), 1.0) + 1e-7, (blk, leaf)
    assert th.step_count() == 1
    th.close()


@pytest.mark.gpu
def test_trainer_trains_segnn_and_runner_mode_all(tmp_path):
    """The reference's trainer is model-agnostic (train/trainer.py:36-89; tests/runner_test.py:14-57 runs train_or_infer end
    to end and expects 0): the Trainer lowers the loss of a SEGNN on the LJ dataset, writes a checkpoint in the reference's
    on-disk format (e3nn leaf names and row order), resumes from it, and `mode: all` of the runner returns 0."""
    import json
    import os
    import shutil
    from lagrangebench_amd.case_setup import case_builder
    from lagrangebench_amd.data import H5Dataset
    from lagrangebench_amd.models import SEGNN, node_irreps
    from lagrangebench_amd.runner import train_or_infer
    from lagrangebench_amd.train import Trainer
    from lagrangebench_amd.utils import load_haiku
    root = os.path.dirname(os.path.abspath(__file__))
    ds_dir = tmp_path / "3D_LJ_3_1214every1"
    shutil.copytree(os.path.join(root, "golden", "3D_LJ_3_1214every1"), ds_dir)
    md = json.load(open(ds_dir / "metadata.json"))
    md.setdefault("write_every", 1)
    json.dump(md, open(ds_dir / "metadata.json", "w"))
    isl, L = 6, 2
    data_train = H5Dataset("train", str(ds_dir), name="lj3d", input_seq_length=isl, extra_seq_length=1)
    data_valid = H5Dataset("valid", str(ds_dir), name="lj3d", input_seq_length=isl, extra_seq_length=10)
    bounds = np.array(md["bounds"])
    case = case_builder(bounds[:, 1] - bounds[:, 0], md, isl, cfg_model={"magnitude_features": True}, noise_std=3e-4)
    irr = node_irreps(md, isl, False, True, True)
    model = SEGNN(irr, "1x1o+1x0e", 64, 1, 1, "1x1o", num_mp_steps=L, n_vels=isl - 1, homogeneous_particles=True)
    cfg_train = {"batch_size": 2, "noise_std": 3e-4,
                 "optimizer": {"lr_start": 1e-3, "lr_final": 1e-5, "lr_decay_rate": 0.1, "lr_decay_steps": 200},
                 "pushforward": {"steps": [-1, 20], "unrolls": [0, 1], "probs": [1, 1]}}
    trainer = Trainer(model, case, data_train, data_valid, cfg_train=cfg_train,
                      cfg_eval={"n_rollout_steps": 10, "train": {"n_trajs": 2, "metrics": ["mse"]}},
                      cfg_logging={"log_steps": 5, "eval_steps": 30}, input_seq_length=isl, seed=0)
    ckp = str(tmp_path / "ckp")
    params, state, opt_state = trainer.train(step_max=60, store_ckp=ckp)
    losses = [l for _, l in trainer.loss_log]
    assert np.isfinite(losses).all() and np.mean(losses[-4:]) < 0.8 * np.mean(losses[:3]), losses
    loaded, _, opt_loaded, step = load_haiku(ckp)
    assert step in (30, 60) and set(opt_loaded) >= {"m", "v", "step"} and np.abs(opt_loaded["v"]).max() > 0
    p2, _, _ = trainer.train(step_max=step + 3, load_ckp=ckp)
    assert set(p2) == set(params)
    cfg = {"mode": "all", "dataset": {"src": str(ds_dir), "name": "lj3d"},
           "model": {"name": "segnn", "num_mp_steps": 1, "input_seq_length": isl, "latent_dim": 64, "magnitude_features": True},
           "train": {"step_max": 12, "batch_size": 1, "pushforward": {"steps": [-1], "unrolls": [0], "probs": [1]}},
           "logging": {"log_steps": 5, "eval_steps": 5, "ckp_dir": str(tmp_path / "ckp2"), "run_name": "r"},
           "eval": {"n_rollout_steps": 5, "train": {"n_trajs": 1, "metrics": ["mse"]},
                    "infer": {"n_trajs": 1, "batch_size": 1, "metrics": ["mse"], "out_type": "none"}}}
    assert train_or_infer(cfg) == 0
