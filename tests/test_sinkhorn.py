"""The `sinkhorn` rollout metric (evaluate/metrics.py:127-136,162-176): oracle pins on CPU, HIP vs
oracle on the GPU.  The optimiser is ott-jax's (third party, not installable here): the oracle
restates its published algorithm and is pinned against closed forms of entropic OT, not against
ott itself ("parity unpinned" for the iteration schedule, see oracle/sinkhorn_oracle.py)."""
import numpy as np
import pytest

from oracle import lb_oracle as O
from oracle import sinkhorn_oracle as SK


def test_oracle_matches_closed_form_2x2():
    """Uniform 2 x 2 entropic OT: P = [[t, 1/2-t], [1/2-t, t]], t/(1/2-t) = exp(-Delta/(2 eps)),
    OT_eps = <P, C> + eps KL(P | a x b)."""
    C = np.array([[0.3, 1.1], [0.9, 0.2]], np.float32)
    a = b = np.array([0.5, 0.5])
    for eps in (0.5, 0.2, 0.07):
        reg, it, err = SK.sinkhorn_solve(C, a, b, eps, threshold=1e-10, max_iterations=20000)
        Cd = C.astype(np.float64)
        delta = Cd[0, 0] + Cd[1, 1] - Cd[0, 1] - Cd[1, 0]
        s = np.exp(-delta / (2 * eps))
        t = 0.5 * s / (1 + s)
        P = np.array([[t, 0.5 - t], [0.5 - t, t]])
        closed = (P * Cd).sum() + eps * (P * np.log(P / 0.25)).sum()
        assert abs(reg - closed) < 1e-6 * abs(closed), (eps, reg, closed)
    # parallel updates with momentum 0.5 (the symmetric-term schedule) reach the same optimum
    reg_p, _, _ = SK.sinkhorn_solve(C, a, b, 0.2, threshold=1e-10, parallel=True, momentum=0.5, max_iterations=20000)
    reg_s, _, _ = SK.sinkhorn_solve(C, a, b, 0.2, threshold=1e-10, max_iterations=20000)
    assert abs(reg_p - reg_s) < 1e-8


def test_oracle_divergence_properties():
    """Single points: S_eps = C(x, y) (periodic displacement, float32 cost); identical clouds: 0;
    symmetric; non-negative; grows with the perturbation."""
    disp, _ = O.space_periodic(np.array([1.0, 1.0]))
    x1, y1 = np.array([[0.05, 0.5]]), np.array([[0.95, 0.5]])
    assert SK.sinkhorn_divergence(disp, x1, y1) == pytest.approx(float(np.float32(0.1 ** 2)), rel=1e-6)
    rng = np.random.default_rng(0)
    x = rng.random((200, 2))
    assert abs(SK.sinkhorn_divergence(disp, x, x)) < 1e-8
    ys = [np.mod(x + s * rng.standard_normal(x.shape), 1.0) for s in (0.005, 0.02)]
    d = [SK.sinkhorn_divergence(disp, x, y) for y in ys]
    assert 0 < d[0] < d[1]
    assert SK.sinkhorn_divergence(disp, ys[0], x) == pytest.approx(d[0], rel=1e-3, abs=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["small2d", "small3d"])
def test_sinkhorn_engine_matches_oracle(name):
    torch = pytest.importorskip("torch")
    from lagrangebench_amd.data import make_case
    from tests._common import hip_case, oracle_case
    ds = make_case(name, n_trajs=2, extra_seq_length=6)
    isl = ds.input_seq_length
    hcase, ocase = hip_case(ds), oracle_case(ds)
    pos = np.stack([ds[0][0], ds[1][0]]).astype(np.float64)     # (B, N, T, dim)
    roll = np.transpose(pos, (0, 2, 1, 3))                       # (B, T, N, dim)
    rng = np.random.default_rng(1)
    dx = float(ds.metadata["dx"])
    target = roll[:, isl:]
    pred = target + 0.05 * dx * rng.standard_normal(target.shape) * np.arange(1, target.shape[1] + 1)[None, :, None, None]
    eng = hcase.engine(2)
    stride = 2
    out, iters = eng.sinkhorn(torch.from_numpy(pred), torch.from_numpy(target), stride, return_iters=True)
    out = out.cpu().numpy()
    assert out.shape == (2, 3)
    for b in range(2):
        for k, t in enumerate(range(0, target.shape[1], stride)):
            d, info = SK.sinkhorn_divergence(ocase.displacement, pred[b, t], target[b, t], return_info=True)
            assert tuple(iters[b, k]) == info["iters"], (b, k, iters[b, k], info["iters"])
            assert abs(out[b, k] - d) <= 1e-9 * info["reg"][0] + 1e-6 * abs(d), (out[b, k], d)
    # identical clouds -> 0 (up to the convergence threshold)
    z = eng.sinkhorn(torch.from_numpy(target), torch.from_numpy(target), 3).cpu().numpy()
    assert np.abs(z).max() < 1e-8


@pytest.mark.gpu
def test_default_infer_metrics_run_including_sinkhorn():
    """defaults.py:143 `eval.infer.metrics = ["mse", "e_kin", "sinkhorn"]`: a default-config
    inference must produce all three (round-1 VERDICT "missing" item 1)."""
    torch = pytest.importorskip("torch")
    from lagrangebench_amd.data import make_case
    from lagrangebench_amd.defaults import defaults
    from lagrangebench_amd.evaluate import averaged_metrics, infer
    from lagrangebench_amd.models import GNS
    from tests._common import hip_case, make_params, oracle_case
    assert list(defaults.eval.infer.metrics) == ["mse", "e_kin", "sinkhorn"]
    n_steps, L = 4, 2
    ds = make_case("small2d", n_trajs=2, extra_seq_length=n_steps)
    ds.metadata.setdefault("dt", 1.0)
    ds.metadata.setdefault("write_every", 1)
    params = make_params(ds, num_mp_steps=L)
    model = GNS(2, 128, 2, L, 16)
    hcase = hip_case(ds)
    out = infer(model, hcase, ds, params=params, cfg_eval_infer={"metrics_stride": 2}, n_rollout_steps=n_steps)
    m = out["rollout_1"]
    assert set(m) >= {"mse", "e_kin", "sinkhorn", "mse1"}
    sk = m["sinkhorn"].cpu().numpy()
    assert sk.shape == (2,) and np.isfinite(sk).all() and (sk > -1e-9).all()
    avg = averaged_metrics(out)
    assert {"val/loss", "val/e_kin", "val/sinkhorn", "val/stdsinkhorn"} <= set(avg)
    # sinkhorn of the device rollout against the oracle's on the same predicted positions
    eng = hcase.engine(1)
    eng.set_particle_type(ds[1][1][None])
    pred, _ = eng.rollout(model.handle(eng, params), ds[1][0][None].astype(np.float64), n_steps)
    pred = pred.cpu().numpy()[0]
    tgt = np.transpose(ds[1][0][:, ds.input_seq_length:], (1, 0, 2)).astype(np.float64)
    want = SK.sinkhorn_rollout(oracle_case(ds).displacement, pred, tgt, 2)
    assert np.allclose(sk, want, rtol=1e-5, atol=1e-12)


@pytest.mark.gpu
def test_sinkhorn_full_size_properties():
    """TGV3D-8k (BASELINE configs[2], 8000 x 8000 periodic cost per frame - far beyond what the NumPy oracle
    finishes in seconds): the defining properties of the divergence on the device path at full size - zero
    on identical clouds, symmetric in its arguments, positive and growing with the perturbation, invariant
    under a relabelling of the particles of one cloud (the transport problem does not see particle ids)."""
    torch = pytest.importorskip("torch")
    from lagrangebench_amd.data import make_case
    from tests._common import hip_case
    ds = make_case("tgv3d", n_trajs=1, extra_seq_length=2)
    isl = ds.input_seq_length
    pos = ds[0][0].astype(np.float64)                            # (N, T, dim)
    N = pos.shape[0]
    assert N >= 8000
    box = np.asarray(ds.box, np.float64)
    dx = float(ds.metadata["dx"])
    rng = np.random.default_rng(5)
    x = pos[:, isl]                                              # one frame
    noise = rng.standard_normal(x.shape)
    y1 = np.mod(x + 0.05 * dx * noise, box)
    y2 = np.mod(x + 0.20 * dx * noise, box)
    eng = hip_case(ds).engine(1)

    def div(a, b):
        return float(eng.sinkhorn(torch.from_numpy(a[None, None]), torch.from_numpy(b[None, None]), 1).cpu().numpy()[0, 0])
    d_xx, d_1, d_1r, d_2 = div(x, x), div(y1, x), div(x, y1), div(y2, x)
    assert abs(d_xx) < 1e-8
    assert d_1 > 0 and d_2 > 4 * d_1                             # ~ quadratic in the displacement
    assert abs(d_1 - d_1r) <= 1e-3 * d_1                         # symmetric up to the convergence threshold
    perm = rng.permutation(N)
    assert abs(div(y1[perm], x) - d_1) <= 1e-6 * d_1


# ---- ot_backend="pot" (metrics.py:178-196) -----------------------------------------------------------------------
def test_pot_oracle_matches_closed_form_and_float32_run():
    """Sinkhorn-Knopp restatement: (1) uniform 2 x 2 problem -> the closed-form plan's <P, C>; (2) the same iteration
    run in float32 (what POT does on the reference's float32 inputs) lands within 2e-5 relative of the float64
    restatement - the tolerance the GPU test quotes against the reference."""
    from oracle import sinkhorn_pot_oracle as PK
    C = np.array([[0.3, 1.1], [0.9, 0.2]], np.float32)
    a = b = np.array([0.5, 0.5])
    for reg in (0.5, 0.3):
        val, it, how = PK.sinkhorn2(a, b, C, reg=reg, numItermax=5000, stopThr=1e-11)
        Cd = C.astype(np.float64)
        delta = Cd[0, 0] + Cd[1, 1] - Cd[0, 1] - Cd[1, 0]
        s = np.exp(-delta / (2 * reg))
        t = 0.5 * s / (1 + s)
        P = np.array([[t, 0.5 - t], [0.5 - t, t]])
        assert how == 1 and abs(val - (P * Cd).sum()) < 1e-8, (reg, val, (P * Cd).sum(), it)
    rng = np.random.default_rng(3)
    disp, _ = O.space_periodic(np.array([1.0, 1.0]))
    x = rng.random((300, 2))
    y = np.mod(x + 0.02 * rng.standard_normal(x.shape), 1.0)
    M = SK.distance_matrix(disp, x, y)
    a = b = np.ones(300) / 300
    v64, it64, how64 = PK.sinkhorn2(a, b, M)
    # float32 re-run of the same loop
    a32, M32 = a.astype(np.float32), M
    u, v = a32.copy(), a32.copy()
    K = np.exp(M32 / np.float32(-0.1))
    Kp = (np.float32(1) / a32)[:, None] * K
    for ii in range(500):
        v = a32 / (K.T @ u)
        u = np.float32(1) / (Kp @ v)
        if ii % 10 == 0 and np.linalg.norm(np.einsum("i,ij,j->j", u, K, v) - a32) < 1e-5:
            break
    v32 = float((u[:, None] * K * v[None, :] * M32).sum())
    assert how64 == 1 and it64 == ii + 1
    assert abs(v32 - v64) < 2e-5 * abs(v64), (v32, v64)
    # the divergence: zero for identical clouds, positive and growing with the perturbation, float32 typed
    d0 = PK.sinkhorn_divergence_pot(disp, x, x)
    d1 = PK.sinkhorn_divergence_pot(disp, x, y)
    d2 = PK.sinkhorn_divergence_pot(disp, x, np.mod(x + 0.05 * rng.standard_normal(x.shape), 1.0))
    assert d0.dtype == np.float32 and d0 == 0 and 0 < d1 < d2


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["small2d", "small3d"])
def test_sinkhorn_pot_engine_matches_oracle(name):
    """lb_sinkhorn_pot vs the Sinkhorn-Knopp restatement: same iteration counts and loop exits, each value within
    1e-9 relative (both fp64 on the float32 cost; the reference's float32 run sits within 2e-5, see the CPU test),
    including the numerical-stop branch (K underflows -> previous scalings kept)."""
    torch = pytest.importorskip("torch")
    from oracle import sinkhorn_pot_oracle as PK
    from lagrangebench_amd.data import make_case
    from tests._common import hip_case, oracle_case
    ds = make_case(name, n_trajs=2, extra_seq_length=6)
    isl = ds.input_seq_length
    hcase, ocase = hip_case(ds), oracle_case(ds)
    pos = np.stack([ds[0][0], ds[1][0]]).astype(np.float64)
    roll = np.transpose(pos, (0, 2, 1, 3))
    rng = np.random.default_rng(1)
    dx = float(ds.metadata["dx"])
    target = roll[:, isl:]
    pred = target + 0.05 * dx * rng.standard_normal(target.shape) * np.arange(1, target.shape[1] + 1)[None, :, None, None]
    eng = hcase.engine(2)
    stride = 2
    out, info = eng.sinkhorn_pot(torch.from_numpy(pred), torch.from_numpy(target), stride, return_info=True)
    out = out.cpu().numpy()
    assert out.shape == (2, 3)
    for b in range(2):
        for k, t in enumerate(range(0, target.shape[1], stride)):
            d, oi = PK.sinkhorn_divergence_pot(ocase.displacement, pred[b, t], target[b, t], return_info=True)
            assert tuple(info[b, k, :3]) == oi["iters"] and tuple(info[b, k, 3:]) == oi["how"], (info[b, k], oi)
            scale = float(oi["values"][0])
            assert abs(out[b, k] - float(d)) <= 3e-7 * scale, (b, k, out[b, k], d)   # a float32 ulp of the values
            assert out[b, k] >= 0 and np.float32(out[b, k]) == out[b, k]
    z = eng.sinkhorn_pot(torch.from_numpy(target), torch.from_numpy(target), 3).cpu().numpy()
    assert (z == 0).all()
    # numerical stop: reg so small that every K(x_i, y_j) underflows for a shifted cloud -> K^T u == 0 at ii = 0
    box = np.asarray(ds.metadata["bounds"], np.float64)
    shifted = target[:, :1] + 0.37 * (box[:, 1] - box[:, 0])
    o2, i2 = eng.sinkhorn_pot(torch.from_numpy(shifted), torch.from_numpy(target[:, :1]), 1, reg=1e-6, return_info=True)
    val, it, how = PK.sinkhorn2(np.ones(target.shape[2]) / target.shape[2], np.ones(target.shape[2]) / target.shape[2],
                                SK.distance_matrix(ocase.displacement, shifted[0, 0], target[0, 0]), reg=1e-6)
    assert how == 2 and it == 1 and i2[0, 0, 0] == 1 and i2[0, 0, 3] == 2


@pytest.mark.gpu
def test_metrics_computer_pot_backend():
    torch = pytest.importorskip("torch")
    from lagrangebench_amd.data import make_case
    from lagrangebench_amd.evaluate import MetricsComputer
    from tests._common import hip_case
    ds = make_case("small2d", n_trajs=1, extra_seq_length=4)
    hcase = hip_case(ds)
    pos = np.transpose(ds[0][0].astype(np.float64), (1, 0, 2))[ds.input_seq_length:]
    rng = np.random.default_rng(0)
    pred = pos + 0.02 * float(ds.metadata["dx"]) * rng.standard_normal(pos.shape)
    mc = MetricsComputer(["sinkhorn"], hcase.displacement, ds.metadata, ds.input_seq_length, stride=2,
                         ot_backend="pot", case=hcase)
    sk = mc(pred, pos)["sinkhorn"]
    assert sk.dtype == torch.float32 and sk.shape == (2,) and bool((sk >= 0).all())
