"""SEGNN beyond the shipped switches (SURVEY.md section 8 row a21's remainder): lmax_hidden / lmax_attributes up to 2
(models/segnn.py:365-400,481-484) and segnn_norm = "instance" / "batch" (models/segnn.py:303,346-351).

CPU: the general-irreps oracle (oracle/segnn_irreps_oracle.py) - 3j symbol properties, O(3) equivariance (the reference's own
SEGNN test, tests/models_test.py:70-87), agreement with the lmax-1 oracle; host logic of models.SEGNN.
GPU: csrc/lb_segnn_gen.hip against that oracle, 1e-5 of the largest entry (exact-fp32 MFMA vs NumPy fp32), every hidden state.
Parity with e3nn-jax itself is unpinned (oracle header, A1 - A10)."""
import numpy as np
import pytest
import torch

from oracle import segnn_irreps_oracle as G
from oracle import segnn_oracle as S
from tests._common import hip_case, oracle_case, rel_err
from tests.test_segnn import _random_graph_features


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


# ------------------------------------------------------------------------------- CPU: 3j symbols, harmonics
def test_w3j_known_values_and_symmetries():
    assert np.allclose(G.w3j(1, 1, 0)[:, :, 0], np.eye(3) / np.sqrt(3))
    eps = np.zeros((3, 3, 3))
    for i, j, k in [(0, 1, 2), (1, 2, 0), (2, 0, 1)]:
        eps[i, j, k], eps[j, i, k] = 1, -1
    assert np.allclose(G.w3j(1, 1, 1), eps / np.sqrt(6))       # the cross product, x y z right-handed
    for l1 in range(3):
        for l2 in range(3):
            for l3 in range(abs(l1 - l2), min(l1 + l2, 2) + 1):
                C = G.w3j(l1, l2, l3)
                assert np.isclose(np.linalg.norm(C), 1.0)
                sign = (-1) ** (l1 + l2 + l3)
                assert np.allclose(C, sign * G.w3j(l2, l1, l3).transpose(1, 0, 2), atol=1e-12)   # 3j column swap
                assert np.allclose(C, G.w3j(l2, l3, l1).transpose(2, 0, 1), atol=1e-12)          # cyclic
                # orthogonality of the coupled basis: sum_{m1 m2} C C = delta / (2 l3 + 1)
                assert np.allclose(np.einsum("ijk,ijl->kl", C, C), np.eye(2 * l3 + 1) / (2 * l3 + 1), atol=1e-12)


def test_spherical_harmonics_l2_match_the_3j_symbols():
    """A9 against A7: Y2(u) is a positive multiple of the 1o x 1o -> 2e product of u with itself, has the "integral"
    norm, and transforms with the same matrices as the 3j symbols' l = 2 index."""
    rng = np.random.default_rng(0)
    v = rng.standard_normal((64, 3))
    u = v / np.linalg.norm(v, axis=1, keepdims=True)
    with S.precision(np.float64):
        y = G.spherical_harmonics(v, 2)
    t = np.einsum("ri,rj,ijk->rk", u, u, G.w3j(1, 1, 2))
    ratio = y[:, 4:] / t
    assert np.all(ratio > 0) and np.allclose(ratio, ratio[0, 0], rtol=1e-9)
    assert np.allclose((y[:, 4:] ** 2).sum(1), 5 / (4 * np.pi))
    assert np.allclose((y[:, 1:4] ** 2).sum(1), 3 / (4 * np.pi))
    assert np.allclose(G.spherical_harmonics(np.zeros((1, 3)), 2)[0, 1:], 0)


def test_weight_balanced_irreps():
    assert G.weight_balanced_chunks(64, 1, 1) == [(32, 0), (32, 1)]            # every shipped config
    assert G.weight_balanced_chunks(64, 2, 2) == [(20, 0), (20, 1), (20, 2)]    # 11 paths: 11 n^2 >= 4096
    assert G.weight_balanced_chunks(64, 2, 1) == [(29, 0), (29, 1)]            # 5 paths
    assert G.weight_balanced_chunks(64, 1, 0) == [(64, 0)]                     # scalar hidden features: one path
    from lagrangebench_amd.models.segnn import weight_balanced_hidden
    for la in range(3):
        for lh in range(3):
            assert weight_balanced_hidden(64, lh, la) == G.weight_balanced_chunks(64, la, lh)[0][0]


def _lmax1_params_from_generic(pg, xn, L, B):
    from lagrangebench_amd.utils import segnn_row_order
    hid = pg["hidden"]
    p1 = {"hidden": hid[0][0], "blocks": B, "layers": L}
    for name, xin, out, _ in G.block_list(xn, hid, L, B):
        if name == "embedding_nodes":
            ops = [xn]
        elif name.endswith("message_0"):
            ops = [hid, hid, G.MSG_CHUNKS]
        elif name.endswith("update_0"):
            ops = [hid, hid]
        else:
            ops = [hid]
        perm = segnn_row_order(ops)
        blk = pg[name]
        ws = blk["w0"][perm] if "w0" in blk else np.zeros((len(perm), 0), np.float32)
        p1[name] = {"ws": ws, "wv": blk["w1"][perm], "b": blk["b"]}
    return p1


def test_general_oracle_equals_lmax1_oracle():
    n, K = 40, 5
    pt = np.random.default_rng(2).integers(0, 3, n)
    xn = G.node_chunks(K, True, True, True, False)
    pg = G.segnn_init(np.random.default_rng(0), xn, num_mp_steps=3, random_bias=True)
    p1 = _lmax1_params_from_generic(pg, xn, 3, 2)
    f = _random_graph_features(np.eye(3))
    o1 = S.segnn_apply(p1, dict(f), pt, K, False)["acc"]
    og = G.segnn_apply(pg, dict(f), pt, K, False)["acc"]
    assert np.abs(o1).max() > 1e-3
    assert np.abs(o1 - og).max() < 2e-6 * np.abs(o1).max()


@pytest.mark.parametrize("lh,la,norm", [(2, 2, None), (2, 1, None), (1, 2, "batch"), (2, 2, "instance"), (1, 1, "batch"),
                                        (0, 1, None)])
def test_general_oracle_is_o3_equivariant(lh, la, norm):
    from scipy.stats import ortho_group
    n, K = 40, 5
    pt = np.random.default_rng(2).integers(0, 3, n)
    xn = G.node_chunks(K, True, True, True, False)
    p = G.segnn_init(np.random.default_rng(0), xn, num_mp_steps=2, lmax_hidden=lh, lmax_attr=la, norm=norm, random_bias=True)
    out0 = G.segnn_apply(p, _random_graph_features(np.eye(3)), pt, K, False)["acc"]
    assert np.abs(out0).max() > 1e-3
    dets = []
    for seed in range(4):
        R = ortho_group.rvs(3, random_state=seed)
        dets.append(np.sign(np.linalg.det(R)))
        out1 = G.segnn_apply(p, _random_graph_features(R), pt, K, False)["acc"]
        assert np.abs(out1 - out0 @ R.T).max() < 5e-6 * max(1.0, np.abs(out0).max())
    assert -1 in dets and 1 in dets


def test_batch_norm_restatement():
    """A10: after the "batch" normalisation every scalar channel has mean bias and every channel's component-mean square is
    weight^2 (up to eps); "instance" on an (N, dim) array returns the bias for scalars and unit-norm irreps."""
    rng = np.random.default_rng(0)
    ch = [(6, 0), (4, 1), (3, 2)]
    x = (rng.standard_normal((500, G.dim_of(ch))) * 3 + 1).astype(np.float32)
    w = rng.uniform(0.5, 2, 13).astype(np.float32)
    b = rng.standard_normal(6).astype(np.float32)
    y = G.batch_norm(x, ch, w, b, False, 1e-5)
    assert np.allclose(y[:, :6].mean(0), b, atol=1e-5)
    assert np.allclose(((y[:, :6] - b) ** 2).mean(0), w[:6] ** 2, rtol=1e-3)
    v = y[:, 6:18].reshape(500, 4, 3)
    assert np.allclose((v ** 2).mean((0, 2)), w[6:10] ** 2, rtol=1e-3)
    yi = G.batch_norm(x, ch, w, b, True, 1e-5)
    assert np.allclose(yi[:, :6], b[None])
    t = yi[:, 18:].reshape(500, 3, 5)
    assert np.allclose((t ** 2).mean(2), w[None, 10:] ** 2, rtol=1e-3)


# ------------------------------------------------------------------------------- CPU: host logic
def _model(lh, la, norm, L=2, blocks=2, units=64, K=5, bound=True, force=True, mag=True, homog=False):
    from lagrangebench_amd.models import SEGNN, node_irreps
    md = {"periodic_boundary_conditions": [not bound, not bound, not bound]}
    irr = node_irreps(md, K + 1, force, mag, homog)
    return SEGNN(irr, "1x1o+1x0e", units, lh, la, "1x1o", num_mp_steps=L, n_vels=K, homogeneous_particles=homog, norm=norm,
                 blocks_per_step=blocks)


def test_model_leaves_match_oracle_and_roundtrip():
    from lagrangebench_amd.utils import segnn_params_from_haiku, segnn_params_to_haiku
    for lh, la, norm in [(2, 2, None), (1, 2, "batch"), (2, 1, "instance"), (1, 1, "batch"), (0, 2, None)]:
        m = _model(lh, la, norm)
        assert m.generic
        xn = G.node_chunks(5, True, True, True, False)
        p = G.segnn_init(np.random.default_rng(0), xn, num_mp_steps=2, lmax_hidden=lh, lmax_attr=la, norm=norm, random_bias=True)
        for blk, leaf, shape in m.gen_leaves():
            assert p[blk][leaf].shape == shape, (blk, leaf)
        n_oracle = sum(v.size for k, blk in p.items() if isinstance(blk, dict) for v in blk.values())
        blob = m.flatten(p)
        assert blob.size == n_oracle
        back = m.unflatten(blob)
        for blk, leaf, _ in m.gen_leaves():
            assert np.array_equal(back[blk][leaf], p[blk][leaf])
        hk = segnn_params_to_haiku(p, m)
        again = segnn_params_from_haiku(hk, m)
        for blk, leaf, _ in m.gen_leaves():
            assert np.array_equal(again[blk][leaf], p[blk][leaf]), (blk, leaf)
    assert not _model(1, 1, None).generic                     # the shipped configuration keeps the fused kernels
    assert _model(1, 1, None, units=32).generic               # hidden 16x0e+16x1o: general path
    with pytest.raises(NotImplementedError):
        _model(3, 1, None)
    with pytest.raises(AssertionError):
        _model(1, 1, "layer")


# ------------------------------------------------------------------------------- GPU
def _gpu_setup(name, scale, L, lh, la, norm, blocks=2, units=64, seed=7, velocity_aggregate="avg"):
    from lagrangebench_amd.data import make_case
    from lagrangebench_amd.models import SEGNN, node_irreps
    ds = make_case(name, n_trajs=2, extra_seq_length=6, scale=scale)
    ds.magnitude_features = True
    isl = ds.input_seq_length
    homog = bool(np.all(ds[0][1] == 0))
    has_force = ds.external_force_fn is not None
    irr = node_irreps(ds.metadata, isl, has_force, True, homog)
    model = SEGNN(irr, "1x1o+1x0e", units, lh, la, "1x1o", num_mp_steps=L, n_vels=isl - 1, homogeneous_particles=homog, norm=norm,
                  blocks_per_step=blocks, velocity_aggregate=velocity_aggregate)
    xn = G.node_chunks(isl - 1, not any(ds.metadata["periodic_boundary_conditions"]), has_force, True, homog)
    assert xn == model._node_chunks
    params = G.segnn_init(np.random.default_rng(seed), xn, num_mp_steps=L, scalar_units=units, lmax_hidden=lh, lmax_attr=la,
                          blocks_per_step=blocks, norm=norm, random_bias=True)
    return ds, model, params, homog


@pytest.mark.gpu
@pytest.mark.parametrize("name,scale,L,lh,la,norm,blocks,units", [
    ("small3d", 1.0, 3, 2, 2, None, 2, 64),        # 20x0e+20x1o+20x2e, attributes up to 2e
    ("dam2d", 0.3, 2, 2, 1, None, 2, 64),          # 25x(0e+1o+2e), walls + external force + particle types
    ("small2d", 1.0, 2, 1, 2, "batch", 2, 64),     # 29x0e+29x1o, BatchNorm on messages and nodes
    ("ldc3d", 0.5, 2, 2, 2, "instance", 2, 64),
    ("small3d", 1.0, 2, 1, 1, "batch", 2, 64),     # the shipped irreps + norm
    ("small3d", 1.0, 2, 1, 1, None, 3, 32),        # hidden 16x0e+16x1o, three blocks per step
    ("small2d", 1.0, 2, 0, 1, None, 1, 64),        # scalar hidden features, one block per step
])
def test_general_segnn_forward_parity(name, scale, L, lh, la, norm, blocks, units):
    _need_gpu()
    ds, model, params, homog = _gpu_setup(name, scale, L, lh, la, norm, blocks, units)
    assert model.generic
    ocase, hcase = oracle_case(ds), hip_case(ds)
    isl = ds.input_seq_length
    pos = np.stack([ds[0][0], ds[1][0]])
    pt = np.stack([ds[0][1], ds[1][1]])
    feats, _ = hcase.allocate_eval((pos[:, :, :isl], pt))
    handle = model.handle(feats.engine, params)
    tap = handle.set_tap(True)
    pred, _ = model.apply(params, {}, (feats, pt))
    acc, tap = _np(pred["acc"]), _np(tap)
    N = pos.shape[1]
    hdim = G.dim_of(params["hidden"])
    assert tap.shape[2] == (hdim + 3) // 4 * 4
    for b in range(2):
        of, _ = ocase.allocate_eval((pos[b][:, :isl].astype(np.float64), pt[b]))
        ref, lat = G.segnn_apply(params, dict(of), pt[b], isl - 1, homog, return_latents=True, norm_eps=model.norm_eps)
        # Yardstick for the norm switches: the same restatement in float64.  e3nn's BatchNorm divides by the batch's (or, for
        # "instance", the single node's) root-mean-square, which amplifies rounding: measured on these cases the float32
        # restatement itself sits 1e-5 .. 3e-4 off its float64 twin and the device (exact-fp32 MFMA, ordered sums) 7e-6 .. 5e-5.
        # Bar: 1e-5 against the float32 oracle without norm; with norm, no worse than 2x the float32 oracle's own distance
        # from float64.
        with S.precision(np.float64):
            ref64, lat64 = G.segnn_apply(params, dict(of), pt[b], isl - 1, homog, return_latents=True, norm_eps=model.norm_eps)
        assert ref64["acc"].dtype == np.float64
        for k, f in enumerate(lat):
            got = tap[k][b * N:(b + 1) * N]
            assert not got[:, hdim:].any()
            if norm is None:
                assert rel_err(got[:, :hdim], f) < 1e-5, f"hidden state {k}"
            else:
                e_dev, e_f32 = rel_err(got[:, :hdim], lat64[k]), rel_err(f, lat64[k])
                assert e_dev <= max(1e-5, 2.0 * e_f32), (k, e_dev, e_f32)
        assert np.abs(ref["acc"]).max() > 1e-4
        if norm is None:
            assert rel_err(acc[b], ref["acc"]) < 1e-5
        else:
            e_dev, e_f32 = rel_err(acc[b], ref64["acc"]), rel_err(ref["acc"], ref64["acc"])
            print(f"[general segnn {name} norm={norm} b={b}] acc: device vs f64 {e_dev:.2e}, f32 oracle vs f64 {e_f32:.2e}")
            assert e_dev <= max(1e-5, 2.0 * e_f32), (e_dev, e_f32)
            assert e_dev < 1e-4
    # no atomics, fixed summation order: bit-reproducible
    again = _np(model.apply(params, {}, (feats, pt))[0]["acc"])
    assert np.array_equal(again, acc)
    handle.set_tap(False)


@pytest.mark.gpu
def test_general_segnn_velocity_last_and_two_models_on_one_engine():
    """velocity_aggregate="last" (segnn.py:530-536) feeds the l <= 2 node attributes; and two general models of different LDS
    footprints alive on one engine (the kernel's dynamic-LDS attribute is per kernel, not per model)."""
    _need_gpu()
    ds, model, params, homog = _gpu_setup("small3d", 1.0, 2, 2, 2, None, velocity_aggregate="last")
    _, small, sparams, _ = _gpu_setup("small3d", 1.0, 1, 2, 1, None, units=16)
    ocase, hcase = oracle_case(ds), hip_case(ds)
    isl = ds.input_seq_length
    pos, pt = ds[0]
    feats, _ = hcase.allocate_eval((pos[None, :, :isl], pt[None]))
    h_big = model.handle(feats.engine, params)
    h_small = small.handle(feats.engine, sparams)       # created second, needs less LDS
    of, _ = ocase.allocate_eval((pos[:, :isl].astype(np.float64), pt))
    for m, p, agg in ((model, params, "last"), (small, sparams, "avg"), (model, params, "last")):
        acc = _np(m.apply(p, {}, (feats, pt[None]))[0]["acc"])[0]
        ref = G.segnn_apply(p, dict(of), pt, isl - 1, homog, velocity_aggregate=agg)["acc"]
        assert rel_err(acc, ref) < 1e-5
    with S.precision(np.float64):
        avg = G.segnn_apply(params, dict(of), pt, isl - 1, homog, velocity_aggregate="avg")["acc"]
    assert rel_err(avg, ref) > 1e-3      # the switch matters on this input
    del h_big, h_small


@pytest.mark.gpu
def test_general_segnn_float32_geometry_and_batch_consistency():
    """dtype=float32 geometry (case.py:169) feeding the general path - DAM2D's walls, external force and particle types at
    lmax 2 - against the oracle run in float32; and a batch of two trajectories gives each trajectory exactly what it gets
    alone (the BatchNorm statistics are per trajectory)."""
    _need_gpu()
    ds, model, params, homog = _gpu_setup("dam2d", 0.3, 2, 2, 2, "batch")
    isl = ds.input_seq_length
    ocase, hcase = oracle_case(ds, dtype=np.float32), hip_case(ds, dtype="float32")
    pos = np.stack([ds[0][0], ds[1][0]])
    pt = np.stack([ds[0][1], ds[1][1]])
    feats, _ = hcase.allocate_eval((pos[:, :, :isl], pt))
    both = _np(model.apply(params, {}, (feats, pt))[0]["acc"])
    for b in range(2):
        of, _ = ocase.allocate_eval((pos[b][:, :isl].astype(np.float32), pt[b]))
        with S.precision(np.float64):
            ref64 = G.segnn_apply(params, dict(of), pt[b], isl - 1, homog, norm_eps=model.norm_eps)["acc"]
        ref32 = G.segnn_apply(params, dict(of), pt[b], isl - 1, homog, norm_eps=model.norm_eps)["acc"]
        e_dev, e_f32 = rel_err(both[b], ref64), rel_err(ref32, ref64)
        assert e_dev <= max(1e-5, 2.0 * e_f32) and e_dev < 2e-4, (b, e_dev, e_f32)
    hcase1 = hip_case(ds, dtype="float32")
    f1, _ = hcase1.allocate_eval((pos[1:2, :, :isl], pt[1:2]))
    alone = _np(model.apply(params, {}, (f1, pt[1:2]))[0]["acc"])[0]
    assert np.array_equal(alone, both[1])


@pytest.mark.gpu
def test_general_segnn_rollout_matches_oracle():
    """lb_segnn_rollout with the general path as the model (device step loop) against the oracle's eval loop, 5 steps."""
    _need_gpu()
    from functools import partial

    from lagrangebench_amd.evaluate.rollout import _eval_batched_rollout, _forward_eval
    from oracle import lb_oracle as O
    ds, model, params, homog = _gpu_setup("small3d", 1.0, 2, 2, 2, "batch")
    params["output"]["w1"] = (params["output"]["w1"] * 0.02).astype(np.float32)
    ocase, hcase = oracle_case(ds), hip_case(ds)
    isl = ds.input_seq_length
    n_steps = 5
    pos = np.stack([ds[0][0], ds[1][0]])
    pt = np.stack([ds[0][1], ds[1][1]])
    feats, nbrs = hcase.allocate_eval((pos[:, :, :isl], pt))
    fwd = partial(_forward_eval, model_apply=model.apply, case_integrate=hcase.integrate)
    fwd._lb_gns = model
    pred, _, _ = _eval_batched_rollout(fwd, hcase.preprocess_eval, hcase, params, {}, (pos, pt), nbrs,
                                       lambda pred, target: {}, n_steps, isl)
    pred = _np(pred)

    def oracle_apply(p, state, sample):
        f, ptype = sample
        return G.segnn_apply(p, f, ptype, isl - 1, homog, norm_eps=model.norm_eps), state

    for b in range(2):
        _, onbrs = ocase.allocate_eval((pos[b][:, :isl].astype(np.float64), pt[b]))
        ref, _, _ = O.eval_batched_rollout(oracle_apply, ocase, params, {}, (pos[b:b + 1], pt[b:b + 1]), onbrs, n_steps, isl)
        dx = float(ds.metadata["dx"]) if "dx" in ds.metadata else 1.0 / 16
        assert np.abs(pred[b] - np.asarray(ref)[0]).max() < 1e-6 * dx
