"""The full-size oracle fixtures (tests/golden/oracle_fullsize/, written by tests/golden/make_oracle_fixtures.py) must
belong to the inputs the GPU tests build TODAY: same synthetic case, same weights.  A stale fixture is not a correctness
problem - the GPU tests fall back to the live oracle - but it silently costs the GPU suite ~6 minutes of its 20-minute
budget, so the CPU suite says so here.  Also: the summaries the tests compare through (64 rows + 2 random projections
per layer) do notice a single corrupted row."""
import os

import numpy as np
import pytest

from tests import _fullsize_oracle as FO


def _stored(name):
    path = os.path.join(FO.FIXDIR, name + ".npz")
    if not os.path.exists(path):
        pytest.fail(f"{path} is missing: run python tests/golden/make_oracle_fixtures.py {name}")
    return np.load(path, allow_pickle=False)


@pytest.mark.parametrize("name", ["rpf2d", "tgv3d", "ldc3d"])
def test_gns_fixtures_match_todays_inputs(name):
    from lagrangebench_amd.data import make_case
    from tests._common import make_params
    L, n_steps = 10, 20
    ds = make_case(name, n_trajs=1, extra_seq_length=n_steps)
    pos, pt = ds[0]
    isl = ds.input_seq_length
    p1 = make_params(ds, num_mp_steps=L, decoder_scale=1.0)
    assert str(_stored(f"gns_fwd_{name}")["input_hash"]) == FO._hash_inputs(pos[:, :isl], pt) + FO.params_hash(p1)
    p2 = make_params(ds, num_mp_steps=L)
    assert str(_stored(f"gns_roll_{name}")["input_hash"]) == FO._hash_inputs(pos, pt) + FO.params_hash(p2)


def test_other_fixtures_match_todays_inputs():
    from lagrangebench_amd.data import make_case
    from lagrangebench_amd.models import SEGNN, node_irreps
    from tests._common import make_params, make_trained_like_params
    ds = make_case("tgv3d", n_trajs=1, extra_seq_length=5)
    pos, pt = ds[0]
    assert str(_stored("gns_pos_tgv3d")["input_hash"]) == FO._hash_inputs(pos, pt) + FO.params_hash(make_params(ds, num_mp_steps=10))
    ds = make_case("tgv2d", n_trajs=2, extra_seq_length=20, scale=1.0)
    pos2 = np.stack([ds[0][0], ds[1][0]])
    assert str(_stored("gns_roll_np_tgv2d_b2")["input_hash"]) == \
        FO._hash_inputs(pos2, ds[0][1]) + FO.params_hash(make_params(ds, num_mp_steps=10))
    ds = make_case("tgv3d", n_trajs=3, extra_seq_length=2)
    isl = ds.input_seq_length
    pos = np.stack([ds[b][0] for b in range(3)])
    pt = np.stack([ds[b][1] for b in range(3)])
    assert str(_stored("gns_fwd_trained_tgv3d_b3")["input_hash"]) == \
        FO._hash_inputs(pos[:, :, :isl], pt) + FO.params_hash(make_trained_like_params(ds, num_mp_steps=10, decoder_scale=1.0))
    ds = make_case("dam2d", n_trajs=1, extra_seq_length=20)
    ds.magnitude_features = True
    isl = ds.input_seq_length
    irr = node_irreps(ds.metadata, isl, ds.external_force_fn is not None, True, False)
    model = SEGNN(irr, "1x1o+1x0e", 64, 1, 1, "1x1o", num_mp_steps=10, n_vels=isl - 1, homogeneous_particles=False)
    params = model.init_params(5)
    params["output"]["wv"] = (params["output"]["wv"] * 0.01).astype(np.float32)
    pos, pt = ds[0]
    assert str(_stored("segnn_roll_dam2d")["input_hash"]) == FO._hash_inputs(pos, pt) + FO.params_hash(params)


def test_layer_summary_notices_one_bad_row():
    """check_layer: an exact copy passes; ONE row (not among the 64 stored ones) off by 1e-3 of the layer's maximum in
    every entry is caught by the projections; a row off by 1e-6 of the maximum (inside the 1e-5 bar) is not."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((8000, 128)).astype(np.float32)
    rows, proj, mx = FO.summarise_layer(x)
    fix = {"rows_0_0": rows, "proj_0_0": proj, "max_0_0": np.asarray(mx)}
    e_rows, e_proj, bar = FO.check_layer(x, fix, 0, 0)
    assert e_rows == 0.0 and e_proj < 1e-2 * bar   # (projections are stored in fp32)
    victim = next(i for i in range(8000) if i not in set(FO.row_selection(8000).tolist()))
    y = x.copy()
    y[victim] += 1e-3 * mx * np.sign(rng.standard_normal(128)).astype(np.float32)
    e_rows, e_proj, bar = FO.check_layer(y, fix, 0, 0)
    assert e_rows == 0.0 and e_proj > bar
    z = x.copy()
    z[victim] += 1e-6 * mx
    e_rows, e_proj, bar = FO.check_layer(z, fix, 0, 0)
    assert e_proj < bar
