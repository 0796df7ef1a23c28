"""CPU tests of host-side logic: weight flattening/packing order, synthetic data, defaults merge,
metric bookkeeping, pytree helpers."""
import numpy as np
import pytest

from oracle import lb_oracle as O


def test_params_layout_matches_oracle_init_and_blob_length():
    from lagrangebench_amd.models.gns import GNS, layer_names
    m = GNS(3, 128, 2, 10, 16)
    p = m.init_params(7, node_in=15, edge_in=4)
    po = O.gns_init(np.random.default_rng(7), node_in=15, edge_in=4, particle_dimension=3)
    assert sorted(p) == sorted(po)
    for k in p:
        for kk in p[k]:
            assert np.array_equal(p[k][kk], po[k][kk]), (k, kk)
    blob = m.flatten(p)
    D = 128
    n = 9 * 16 + (31 * D + D + D * D + D + 2 * D) + (4 * D + D + D * D + D + 2 * D)
    n += 10 * ((3 * D * D + D + D * D + D + 2 * D) + (2 * D * D + D + D * D + D + 2 * D))
    n += D * D + D + D * 3 + 3
    assert blob.size == n == sum(v.size for d in p.values() for v in d.values())
    # 1.2M parameters (docs/pages/baselines.rst:62)
    assert 1.20e6 < n < 1.22e6
    assert layer_names(2) == ["enc_node", "enc_edge", "proc0_edge", "proc0_node", "proc1_edge", "proc1_node",
                              "decoder"]


def test_synthetic_cases_have_reference_geometry():
    from lagrangebench_amd.data import make_case
    ds = make_case("tgv2d", n_trajs=1, extra_seq_length=2)
    pos, pt = ds[0]
    assert pos.shape == (2500, 8, 2) and pos.dtype == np.float32 and pt.shape == (2500,)
    assert abs(ds.metadata["default_connectivity_radius"] - 0.029) < 1e-12
    # ~6.7 edges per particle incl. self, E_cap ~ 21k at multiplier 1.25 (tutorial.ipynb cell 17)
    disp, _ = O.space_periodic(ds.box)
    nl = O.neighbor_list(disp, ds.box, 0.029, 1.25).allocate(pos[:, 5].astype(np.float64))
    assert 6.0 < nl.occupancy / 2500 < 7.5
    assert 19000 < nl.max_occupancy < 23500
    ds3 = make_case("tgv3d", n_trajs=1, extra_seq_length=1)
    assert ds3[0][0].shape == (8000, 7, 3)
    dam = make_case("dam2d", n_trajs=1, extra_seq_length=1)
    assert dam.multiplier == 2.0 and dam.isotropic_norm and dam.external_force_fn is not None
    assert 5000 < dam[0][0].shape[0] < 6500 and set(np.unique(dam[0][1])) == {0, 1}
    ldc = make_case("ldc3d", n_trajs=1, extra_seq_length=1)
    assert set(np.unique(ldc[0][1])) == {0, 1, 2} and 7500 < ldc[0][0].shape[0] < 8500
    # kinematic lid moves, solid walls do not
    p, t = ldc[0]
    assert np.abs(p[t == 1, 0] - p[t == 1, -1]).max() == 0 and np.abs(p[t == 2, 0, 0] - p[t == 2, 1, 0]).min() > 0


def test_defaults_merge_and_metrics_bookkeeping():
    import torch
    from lagrangebench_amd.defaults import defaults, merge
    from lagrangebench_amd.evaluate.metrics import averaged_metrics
    cfg = merge(defaults.eval.infer, {"batch_size": 4})
    assert cfg.batch_size == 4 and cfg.n_extrap_steps == 0 and cfg.metrics == ["mse", "e_kin", "sinkhorn"] and cfg.out_type == "pkl"  # defaults.py:136-150
    assert defaults.neighbors.multiplier == 1.25 and defaults.model.input_seq_length == 6
    m = {"rollout_0": {"mse": torch.tensor([1.0, 3.0])}, "rollout_1": {"mse": torch.tensor([2.0, 2.0])}}
    assert averaged_metrics(m)["val/loss"] == pytest.approx(2.0)


def test_utils_mask_and_broadcast():
    import torch
    from lagrangebench_amd.utils import NodeType, broadcast_from_batch, broadcast_to_batch, get_kinematic_mask
    pt = np.array([0, 1, 2, 3, -1])
    assert (get_kinematic_mask(pt) == O.get_kinematic_mask(pt)).all()
    assert (get_kinematic_mask(torch.tensor(pt)).numpy() == O.get_kinematic_mask(pt)).all()
    assert NodeType.SIZE == 9
    tree = {"a": np.arange(3), "b": (torch.ones(2), np.zeros((2, 2)))}
    b = broadcast_to_batch(tree, 4)
    assert b["a"].shape == (4, 3) and b["b"][0].shape == (4, 2)
    s = broadcast_from_batch(b, 2)
    assert (s["a"] == tree["a"]).all()


def test_pack_weight_layout():
    """The fragment packing of lb_pack_weight, restated: element (kq, mb, lane, j) holds
    W[8kq + 4(lane>>5) + j][32mb + (lane&31)] - checked through a tiny ctypes-free reimplementation
    against the MFMA operand maps documented in DESIGN.md."""
    K, M = 16, 64
    W = np.arange(K * M, dtype=np.float32).reshape(K, M)
    out = np.zeros(K * M, np.float32)
    NKQ, NMB = K // 8, M // 32
    for kq in range(NKQ):
        for mb in range(NMB):
            for lane in range(64):
                for j in range(4):
                    out[((kq * NMB + mb) * 64 + lane) * 4 + j] = W[8 * kq + 4 * (lane >> 5) + j, 32 * mb + (lane & 31)]
    assert sorted(out.tolist()) == sorted(W.ravel().tolist())  # a permutation: nothing dropped
    # lane 37 (row 5, upper half), kq=1, mb=1, j=2 -> k = 8+4+2 = 14, m = 32+5
    assert out[((1 * NMB + 1) * 64 + 37) * 4 + 2] == W[14, 37]


def test_write_vtk_and_pkl2vtk(tmp_path):
    """evaluate/utils.py:9-77 - legacy VTK polydata per frame (ASCII fallback without pyvista)."""
    import pickle
    from lagrangebench_amd.evaluate import pkl2vtk, write_vtk
    rng = np.random.default_rng(0)
    r = rng.random((7, 2))
    write_vtk({"r": r, "tag": np.arange(7, dtype=np.int32), "v": rng.random((7, 2))}, str(tmp_path / "a.vtk"))
    txt = (tmp_path / "a.vtk").read_bytes()
    assert txt.startswith(b"# vtk DataFile") and b"POINTS 7" in txt
    roll = {"predicted_rollout": rng.random((3, 7, 3)), "ground_truth_rollout": rng.random((3, 7, 3)),
            "particle_type": np.zeros(7, np.int32)}
    with open(tmp_path / "rollout_0.pkl", "wb") as f:
        pickle.dump(roll, f)
    pkl2vtk(str(tmp_path / "rollout_0.pkl"), str(tmp_path / "vtk"))
    names = sorted(p.name for p in (tmp_path / "vtk").iterdir())
    assert names == ["rollout_0_0.vtk", "rollout_0_1.vtk", "rollout_0_2.vtk",
                     "rollout_0_ref_0.vtk", "rollout_0_ref_1.vtk", "rollout_0_ref_2.vtk"]


def test_mfma_chain_report_tool_runs():
    """tools/mfma_chain_check.py (profiles/HISTORY.md section 4): compiles a kernel source to gfx950 assembly (no GPU
    needed) and reports the spacing of dependent MFMA pairs."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "mfma_chain_check.py"),
                        os.path.join(root, "lagrangebench_amd", "csrc", "lb_edge16v.hip")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l for l in r.stdout.splitlines() if "k_edge16v" in l]
    counts = [int(l.split(" mfma ")[1].split()[0]) for l in rows]
    assert rows and all(c % 192 == 0 for c in counts), r.stdout[-1500:]  # 2 GEMMs x 96 MFMAs per tile
