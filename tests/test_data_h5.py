"""H5Dataset mirror + ctypes HDF5 reader against the reference's own debugging dataset
(tests/golden/3D_LJ_3_1214every1 = /root/reference/tests/3D_LJ_3_1214every1, data files)."""
import json
import os

import numpy as np
import pytest

DS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "3D_LJ_3_1214every1")


def test_h5_reader_matches_h5dump_decoding(golden_dir):
    from lagrangebench_amd.data import h5
    ref = np.load(os.path.join(golden_dir, "lj3d_valid.npz"))
    with h5.open_file(os.path.join(DS, "valid.h5")) as f:
        assert list(f.keys()) == ["00000"]
        d = f["00000/position"]
        assert tuple(d.shape) == (405, 3, 3)
        assert np.array_equal(np.asarray(d[:]), ref["position"])
        assert np.array_equal(np.asarray(d[100:203]), ref["position"][100:203])
        assert np.array_equal(np.asarray(f["00000/particle_type"][:]), ref["particle_type"])


def test_h5dataset_eval_split_like_reference_rollout_test():
    """rollout_test.py:37-46: H5Dataset('valid', isl=3, extra=100) -> 405 // 103 = 3 chunks of 103 frames."""
    from lagrangebench_amd.data import H5Dataset
    ds = H5Dataset("valid", DS, name="lj3d", input_seq_length=3, extra_seq_length=100)
    assert ds.num_samples == 3 and len(ds) == 3 and ds.subseq_length == 103
    pos, pt = ds[0]
    assert pos.shape == (3, 103, 3) and pos.dtype == np.float32 and pt.tolist() == [0, 0, 0]
    pos1, _ = ds[1]
    from lagrangebench_amd.data import h5
    with h5.open_file(os.path.join(DS, "valid.h5")) as f:
        full = np.asarray(f["00000/position"][:])
    assert np.array_equal(pos1, full[103:206].transpose(1, 0, 2))
    assert ds.metadata["default_connectivity_radius"] == 3.0 and ds.external_force_fn is None
    with pytest.raises(AssertionError):
        H5Dataset("valid", DS, name="lj3d", input_seq_length=3, extra_seq_length=0)


def test_h5dataset_train_windows():
    from lagrangebench_amd.data import H5Dataset
    ds = H5Dataset("train", DS, name="lj3d", input_seq_length=6, extra_seq_length=2)
    assert ds.subseq_length == 9 and ds.num_samples == 1214 - 9 + 1
    w, pt = ds[7]
    assert w.shape == (3, 9, 3)
    w2, _ = ds[8]
    assert np.array_equal(w[:, 1:], w2[:, :-1])


def test_force_py_is_compiled_to_a_forcespec(tmp_path):
    """force.py of the RPF / DAM datasets (JAX code) -> device ForceSpec by probing."""
    from lagrangebench_amd.data.data import _load_force_fn, force_spec_from_callable, get_dataset_name_from_path
    (tmp_path / "force.py").write_text(
        "import jax.numpy as jnp\n\n"
        "def force_fn(r):\n"
        "    return jnp.where(r[1] > 1.0, jnp.array([-1.0, 0.0]), jnp.array([1.0, 0.0]))\n")
    fn = _load_force_fn(str(tmp_path / "force.py"))
    spec = force_spec_from_callable(fn, [[0.0, 1.0], [0.0, 2.0]])
    assert spec.kind == 1 and spec.axis == 1 and abs(spec.split - 1.0) < 1e-12
    assert tuple(spec.f_lo) == (1.0, 0.0) and tuple(spec.f_hi) == (-1.0, 0.0)
    (tmp_path / "g.py").write_text("import jax.numpy as jnp\n\ndef force_fn(r):\n    return jnp.array([0.0, -1.0])\n")
    spec = force_spec_from_callable(_load_force_fn(str(tmp_path / "g.py")), [[0.0, 5.486], [0.0, 2.12]])
    assert spec.kind == 1 and tuple(spec.f_lo) == tuple(spec.f_hi) == (0.0, -1.0)
    # a band narrower than the probe grid (1/66 of the box) and a force that depends on two coordinates: the probing
    # would compile both to a constant - the random-point verification must send them to the host callable instead
    (tmp_path / "band.py").write_text(
        "import jax.numpy as jnp\n\n"
        "def force_fn(r):\n"
        "    return jnp.where((r[1] > 0.700) & (r[1] < 0.705), jnp.array([5.0, 0.0]), jnp.array([0.0, 0.0]))\n")
    spec = force_spec_from_callable(_load_force_fn(str(tmp_path / "band.py")), [[0.0, 1.0], [0.0, 1.0]])
    assert spec.kind == 2, "narrow band mis-compiled"                      # LB_FORCE_BUFFER: evaluated by the callable
    (tmp_path / "diag.py").write_text(
        "import jax.numpy as jnp\n\n"
        "def force_fn(r):\n"
        "    return jnp.where(r[0] + r[1] > 1.7, jnp.array([0.0, 1.0]), jnp.array([0.0, -1.0]))\n")
    spec = force_spec_from_callable(_load_force_fn(str(tmp_path / "diag.py")), [[0.0, 1.0], [0.0, 1.0]])
    assert spec.kind == 2, "two-coordinate force mis-compiled"
    assert get_dataset_name_from_path("/data/2D_TGV_2500_10kevery100") == "tgv2d"
    assert get_dataset_name_from_path("/data/3D_RPF_8000_10kevery10/") == "rpf3d"
