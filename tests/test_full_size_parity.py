"""BASELINE.json configs [1]-[4] at FULL size, HIP engine vs the CPU oracle (round-1 VERDICT item 1).

For every GNS config: one forward pass with per-layer node-latent taps and a 20-step device rollout,
compared with the oracle (`oracle/lb_oracle.py` geometry + `oracle/lb_oracle_torch.py` network - the
torch-CPU twin of the NumPy network, itself checked against it in tests/test_oracle_golden.py) on the
same seeded inputs / weights.  For the SEGNN config: one full-size forward with per-layer taps.

Tolerances (north_star): edge list bit-exact, per-layer node latents and accelerations within 1e-5
relative (fp32), rollout MSE within 1e-5 absolute and 1e-3 relative of the oracle's.
The oracle is NOT the reference JAX run (JAX cannot be installed here: parity of the network layer
stays "unpinned") - what these tests add is that nothing size-dependent
(tile walks, XCD partitioning, partial-sum slots, capacity growth) breaks parity at the sizes the
benchmark runs.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import lb_oracle as O  # noqa: E402
from oracle import lb_oracle_torch as OT  # noqa: E402
from tests import _fullsize_oracle as FO  # noqa: E402
from tests._common import (elementwise_stats, hip_case, make_params, make_trained_like_params, oracle_case,  # noqa: E402
                           rel_err)


def _np(t):
    return t.detach().cpu().numpy()


def _torch_apply(L):
    cache = {}

    def apply(params, state, sample):
        feats, ptype = sample
        pt = cache.setdefault(id(params), OT.params_to_torch(params))
        return OT.gns_apply(pt, feats, ptype, num_mp_steps=L, skip_padding=True), state
    return apply


# (case, BASELINE config index, steps compared with the oracle)
GNS_CONFIGS = [("rpf2d", 1, 20), ("tgv3d", 2, 20), ("ldc3d", 3, 20)]


@pytest.mark.parametrize("name,cfg_idx,n_steps", GNS_CONFIGS, ids=[c[0] for c in GNS_CONFIGS])
def test_full_size_gns_forward_and_rollout_vs_oracle(name, cfg_idx, n_steps):
    from lagrangebench_amd.data import make_case
    from lagrangebench_amd.evaluate import infer
    from lagrangebench_amd.models import GNS
    L = 10
    ds = make_case(name, n_trajs=1, extra_seq_length=n_steps)
    dim, isl = len(ds.box), ds.input_seq_length
    hcase = hip_case(ds)
    pos, pt = ds[0]
    N = len(pt)
    assert N >= 3000  # full size, not a scaled-down case

    # ---- one forward: edge list bit-exact, every layer's node latents, accelerations.  The oracle's side comes from
    # tests/golden/oracle_fullsize/ when its inputs are the ones built here (tests/_fullsize_oracle.py), else it runs live.
    feats, nbrs = hcase.allocate_eval((pos[:, :isl], pt))
    params = make_params(ds, num_mp_steps=L, decoder_scale=1.0)
    fix, hit = FO.cached(f"gns_fwd_{name}", FO._hash_inputs(pos[:, :isl], pt) + FO.params_hash(params),
                         lambda: FO.gns_forward(ds, params, L))
    print(f"[full size {name}] oracle forward from the {'fixture' if hit else 'LIVE oracle'}")
    ne = int(fix["ne_0"])
    idx = _np(nbrs.idx)
    assert int(_np(nbrs.n_edges)) == ne and FO.edges_digest(idx[:, :ne]) == str(fix["edges_sha_0"]) and (idx[:, ne:] == N).all()
    model = GNS(dim, 128, 2, L, 16)
    handle = model.handle(feats.engine, params)
    tap = handle.set_tap(True)
    acc = _np(model.apply(params, {}, (feats, pt))[0]["acc"])
    tap = _np(tap).copy()
    handle.set_tap(False)
    for k in range(L + 1):   # k = 0: encoder output, k: after layer k - 64 complete rows and every row through 2 projections
        e_rows, e_proj, bar = FO.check_layer(tap[k][:N], fix, 0, k)
        assert e_rows < 1e-5 and e_proj < bar, f"layer {k - 1}: rows {e_rows:.2e}, projections {e_proj:.2e} (bar {bar:.2e})"
    ref_acc, truth = fix["acc_0"], fix["truth_0"]
    assert rel_err(acc, ref_acc) < 1e-5
    # ---- element-wise: north_star asks for 1e-5 relative PER acceleration, the max-norm bar above lets small
    # entries off.  Yardstick = the same network evaluated in fp64; the bar = what a plain fp32 evaluation (the
    # torch-CPU oracle) itself achieves against it.  Entries below 1e-3 of the largest are left out (they are
    # differences of O(1) terms: no fp32 evaluation keeps 1e-5 relative there).
    p999_h, max_h, n_h = elementwise_stats(acc, truth)
    p999_o, max_o, _ = elementwise_stats(ref_acc, truth)
    print(f"[elementwise {name}] engine vs fp64: p99.9 {p999_h:.2e} max {max_h:.2e} | fp32 oracle vs fp64: "
          f"p99.9 {p999_o:.2e} max {max_o:.2e} ({n_h} entries)")
    assert p999_h <= max(3.0 * p999_o, 1e-5), (p999_h, p999_o)
    assert max_h <= max(3.0 * max_o, 1e-4), (max_h, max_o)

    # ---- 20-step rollout (for RPF2D the first 20 of the 400 steps of configs[1])
    p2 = make_params(ds, num_mp_steps=L)
    out = infer(model, hcase, ds, params=p2, cfg_eval_infer={"batch_size": 1, "metrics": ["mse"]},
                n_rollout_steps=n_steps)
    rfix, hit = FO.cached(f"gns_roll_{name}", FO._hash_inputs(pos, pt) + FO.params_hash(p2),
                          lambda: FO.gns_rollout(ds, p2, L, n_steps))
    print(f"[full size {name}] oracle rollout from the {'fixture' if hit else 'LIVE oracle'}")
    mse_h, mse_o = _np(out["rollout_0"]["mse"]), rfix["mse"][0]
    assert mse_h.shape == (n_steps,)
    assert np.abs(mse_h - mse_o).max() <= 1e-5
    assert np.allclose(mse_h, mse_o, rtol=1e-3, atol=1e-12), (mse_h, mse_o)


def test_full_size_gns_positions_track_oracle_tgv3d():
    """Per-step positions of the TGV3D-8k device rollout against the oracle's (a stronger statement
    than the MSE: accelerations within 1e-5 relative => positions within ~1e-5 * acc_std)."""
    from lagrangebench_amd.data import make_case
    from lagrangebench_amd.models import GNS
    L, n_steps = 10, 5
    ds = make_case("tgv3d", n_trajs=1, extra_seq_length=n_steps)
    isl = ds.input_seq_length
    ocase, hcase = oracle_case(ds), hip_case(ds)
    pos, pt = ds[0]
    params = make_params(ds, num_mp_steps=L)
    model = GNS(3, 128, 2, L, 16)
    eng = hcase.engine(1)
    eng.set_particle_type(pt[None])
    pred, _ = eng.rollout(model.handle(eng, params), pos[None].astype(np.float64), n_steps)
    fix, hit = FO.cached("gns_pos_tgv3d", FO._hash_inputs(pos, pt) + FO.params_hash(params),
                         lambda: FO.gns_rollout(ds, params, L, n_steps))
    print(f"[positions tgv3d] oracle from the {'fixture' if hit else 'LIVE oracle'}")
    sel = FO.track_selection(len(pt))   # 1500 sampled particles at every step + the per-step MSE of all of them
    assert np.abs(_np(pred)[0][:, sel] - fix["track"][0]).max() < 1e-6 * float(ds.metadata["dx"])
    truth = np.transpose(pos[:, isl:isl + n_steps], (1, 0, 2)).astype(np.float64)
    d = _np(pred)[0] - truth
    d -= ds.box * np.round(d / ds.box)     # the metric's periodic displacement (metrics.py:139-142)
    mse_h = (d ** 2).mean(axis=(1, 2))
    assert np.allclose(mse_h, fix["mse"][0], rtol=1e-6, atol=1e-14), (mse_h, fix["mse"][0])


def test_full_size_segnn_dam2d_forward_vs_oracle():
    """BASELINE configs[4] (DamBreak2D ~5.5k particles, SEGNN-10-64, free surface) at full size: one
    forward pass, every layer's hidden state and the acceleration against oracle/segnn_oracle.py."""
    from lagrangebench_amd.data import make_case
    from lagrangebench_amd.models import SEGNN, node_irreps
    from oracle import segnn_oracle as S
    L = 10
    ds = make_case("dam2d", n_trajs=1, extra_seq_length=2)
    ds.magnitude_features = True
    isl = ds.input_seq_length
    irr = node_irreps(ds.metadata, isl, ds.external_force_fn is not None, True, False)
    model = SEGNN(irr, "1x1o+1x0e", 64, 1, 1, "1x1o", num_mp_steps=L, n_vels=isl - 1, homogeneous_particles=False)
    params = model.init_params(5)
    ocase, hcase = oracle_case(ds), hip_case(ds)
    pos, pt = ds[0]
    N = len(pt)
    assert N >= 5000
    feats, nbrs = hcase.allocate_eval((pos[:, :isl], pt))
    of, on = ocase.allocate_eval((pos[:, :isl].astype(np.float64), pt))
    want = O.canonical_edges(on.idx, N)
    assert (_np(nbrs.idx)[:, :want.shape[1]] == want).all()
    sh = model.handle(feats.engine, params)
    stap = sh.set_tap(True)
    acc = _np(model.apply(params, {}, (feats, pt))[0]["acc"])
    stap = _np(stap).copy()
    sh.set_tap(False)
    ref, lat = S.segnn_apply(params, of, pt, isl - 1, False, return_latents=True)
    for k, f in enumerate(lat):
        w = np.concatenate([f.s, f.v[:, :, 0], f.v[:, :, 1], f.v[:, :, 2]], axis=1)
        assert rel_err(stap[k][:N], w) < 1e-5, f"layer {k}"
    hid = float(np.abs(lat[-1].s).max())
    assert np.abs(acc - ref["acc"]).max() < 1e-5 * max(hid, float(np.abs(ref["acc"]).max()))


def test_full_size_segnn_dam2d_rollout_vs_oracle():
    """BASELINE configs[4] at full size, the ROLLOUT: 20 steps of lb_segnn_rollout (device step
    loop: single-launch cell binning, search, k_sg_embed, 10 x (k_sg_msg + k_sg_upd), k_sg_readout with the integrator
    in its epilogue) against the oracle's eval loop (NumPy neighbor list + oracle/segnn_oracle.py) on the same
    trajectory and weights: every step's positions within 1e-6 dx, the 20-step MSE within 1e-5."""
    import time
    from lagrangebench_amd.data import make_case
    from lagrangebench_amd.models import SEGNN, node_irreps
    from oracle import segnn_oracle as S
    L, n_steps = 10, 20
    ds = make_case("dam2d", n_trajs=1, extra_seq_length=n_steps)
    ds.magnitude_features = True
    isl = ds.input_seq_length
    irr = node_irreps(ds.metadata, isl, ds.external_force_fn is not None, True, False)
    model = SEGNN(irr, "1x1o+1x0e", 64, 1, 1, "1x1o", num_mp_steps=L, n_vels=isl - 1, homogeneous_particles=False)
    params = model.init_params(5)
    params["output"]["wv"] = (params["output"]["wv"] * 0.01).astype(np.float32)  # accelerations of a physical size
    ocase, hcase = oracle_case(ds), hip_case(ds)
    pos, pt = ds[0]
    N = len(pt)
    assert N >= 5000
    eng = hcase.engine(1)
    eng.set_particle_type(pt[None])
    traj = eng.prepare_traj(pos[None])
    pred = _np(eng.rollout(model.handle(eng, params), traj, n_steps)[0])[0]  # (T, N, dim)

    fix, hit = FO.cached("segnn_roll_dam2d", FO._hash_inputs(pos, pt) + FO.params_hash(params),
                         lambda: FO.segnn_rollout(ds, params, n_steps, isl - 1))
    print(f"[segnn dam2d rollout] oracle from the {'fixture' if hit else 'LIVE oracle'}")
    dx = float(ds.metadata["dx"])
    sel = FO.track_selection(N)
    err = np.abs(pred[:, sel] - fix["track"]).max(axis=(1, 2))   # 1500 sampled particles at every step
    print("[segnn dam2d rollout] max |dpos| / dx per step:", np.array2string(err / dx, precision=2))
    assert err.max() < 1e-6 * dx
    truth = np.transpose(pos[:, isl:isl + n_steps], (1, 0, 2))
    mse_h = ((pred - truth) ** 2).mean(axis=(1, 2))           # ... and all of them through the per-step MSE
    mse_o = fix["mse"]
    assert np.abs(mse_h - mse_o).max() <= 1e-5 and np.allclose(mse_h, mse_o, rtol=1e-3, atol=1e-12)


def test_batched_40k_nodes_forward_vs_oracle():
    """B = 5 TGV3D-8k trajectories = 40 000 nodes in one graph: the size class of the benchmark line
    (lb_node16s single-pass node kernel from 16 k nodes, full 256-workgroup edge walk).  Per-layer
    node latents and accelerations of the first and the last trajectory against the oracle."""
    from lagrangebench_amd.data import make_case
    from lagrangebench_amd.models import GNS
    L, B = 10, 5
    ds = make_case("tgv3d", n_trajs=B, extra_seq_length=2)
    isl = ds.input_seq_length
    ocase, hcase = oracle_case(ds), hip_case(ds)
    pos = np.stack([ds[i][0] for i in range(B)])
    pt = np.stack([ds[i][1] for i in range(B)])
    N = pos.shape[1]
    assert B * N >= 32768
    params = make_params(ds, num_mp_steps=L, decoder_scale=1.0)
    model = GNS(3, 128, 2, L, 16)
    feats, nbrs = hcase.allocate_eval((pos[:, :, :isl], pt))
    handle = model.handle(feats.engine, params)
    tap = handle.set_tap(True)
    acc = _np(model.apply(params, {}, (feats, pt))[0]["acc"])
    tap = _np(tap).copy()
    handle.set_tap(False)
    pt_t = OT.params_to_torch(params)
    for b in (0, B - 1):
        of, on = ocase.allocate_eval((pos[b][:, :isl].astype(np.float64), pt[b]))
        assert (_np(nbrs.idx)[b][:, :int(_np(nbrs.n_edges)[b])] == O.canonical_edges(on.idx, N)).all()
        ref, inter = OT.gns_apply(pt_t, of, pt[b], num_mp_steps=L, skip_padding=True, return_intermediates=True)
        assert rel_err(tap[0][b * N:(b + 1) * N], inter["enc_n"]) < 1e-5
        for k in range(L):
            assert rel_err(tap[k + 1][b * N:(b + 1) * N], inter[f"n{k}"]) < 1e-5, (b, k)
        assert rel_err(acc[b], ref["acc"]) < 1e-5


@pytest.mark.parametrize("name,batch", [("tgv2d", 1), ("small3d", 1), ("tgv3d", 3)], ids=["tgv2d", "small3d", "tgv3d_b3"])
def test_trained_like_weight_statistics(name, batch):
    """Random-init weights are the friendliest case for the fp16 hi/lo split (every operand O(1)).  A trained
    checkpoint is not bound to that: heavy-tailed matrices, LayerNorm scales spread over three decades, biases much
    larger than the weights' scale (tests/_common.py: make_trained_like_params).  The engine - whichever arithmetic its
    range guard ends up in - must stay within 1e-5 of the oracle per layer and as accurate as an fp32 evaluation
    element-wise; batch 3 of TGV3D runs the wave-per-tile kernels, one trajectory the M-split ones."""
    from lagrangebench_amd.data import make_case
    from lagrangebench_amd.models import GNS
    L = 10
    ds = make_case(name, n_trajs=batch, extra_seq_length=2)
    dim, isl = len(ds.box), ds.input_seq_length
    ocase, hcase = oracle_case(ds), hip_case(ds)
    params = make_trained_like_params(ds, num_mp_steps=L, decoder_scale=1.0)
    model = GNS(dim, 128, 2, L, 16)
    pos = np.stack([ds[b][0] for b in range(batch)])
    pt = np.stack([ds[b][1] for b in range(batch)])
    N = pos.shape[1]
    feats, nbrs = hcase.allocate_eval((pos[:, :, :isl], pt))
    handle = model.handle(feats.engine, params)
    tap = handle.set_tap(True)
    acc = _np(model.apply(params, {}, (feats, pt))[0]["acc"])
    tap = _np(tap).copy()
    handle.set_tap(False)
    mode, flags = feats.engine.math_mode()
    print(f"[trained-like {name}] engine arithmetic after the forward: mode {mode} (0 = fp32, 1 = guarded f16x2), "
          f"guard flags {flags}, kernels {feats.engine.kernel_names()}")
    fix, hit = FO.cached(f"gns_fwd_trained_{name}_b{batch}", FO._hash_inputs(pos[:, :, :isl], pt) + FO.params_hash(params),
                         lambda: FO.gns_forward(ds, params, L, traj_ids=tuple(range(batch))))
    print(f"[trained-like {name}] oracle from the {'fixture' if hit else 'LIVE oracle'}")
    for b in range(batch):
        sl = slice(b * N, (b + 1) * N)
        for k in range(L + 1):
            e_rows, e_proj, bar = FO.check_layer(tap[k][sl], fix, b, k)
            assert e_rows < 1e-5 and e_proj < bar, f"traj {b} layer {k - 1}: rows {e_rows:.2e}, projections {e_proj:.2e} (bar {bar:.2e})"
        ref_acc, truth = fix[f"acc_{b}"], fix[f"truth_{b}"]
        assert rel_err(acc[b], ref_acc) < 1e-5
        p999_h, max_h, n_h = elementwise_stats(acc[b], truth)
        p999_o, max_o, _ = elementwise_stats(ref_acc, truth)
        print(f"[trained-like {name} b{b}] engine vs fp64: p99.9 {p999_h:.2e} max {max_h:.2e} | fp32 oracle vs fp64: "
              f"p99.9 {p999_o:.2e} max {max_o:.2e} ({n_h} entries)")
        assert p999_h <= max(3.0 * p999_o, 1e-5), (p999_h, p999_o)
        assert max_h <= max(3.0 * max_o, 1e-4), (max_h, max_o)
