"""H5Dataset: mirror of lagrangebench/data/data.py:33-269.

Same constructor, same windows (`get_window`, train) / trajectories (`get_trajectory`, valid & test,
split into ``sequence_length // (input_seq_length + extra_seq_length)`` chunks), same attributes
(``metadata``, ``input_seq_length``, ``num_samples``, ``external_force_fn``).  Files are read with
``lagrangebench_amd.data.h5`` (ctypes on libhdf5; h5py if present).  Not mirrored: the automatic
Zenodo download (no network here - a missing dataset raises) and the matscipy padding branch.

``force.py`` shipped with the RPF / DAM datasets is JAX code; it is executed against a small
``jax.numpy`` -> NumPy shim (enough for the published force functions) and then *compiled* into a
device-evaluated ``ForceSpec`` by probing it (`force_spec`): the published forces are piecewise
constant along one axis.  Anything else falls back to a host callable.
"""
from __future__ import annotations

import bisect
import importlib.util
import json
import os
import os.path as osp
import re
import sys
import types
import warnings
from typing import Callable, Optional

import numpy as np

from . import h5


def get_dataset_name_from_path(path: str) -> str:
    """data.py:272-298: {2D|3D}_{ABC}_... -> abc2d / abc3d, else the directory name."""
    d = osp.basename(osp.normpath(path))
    m = re.search(r"(?:2D|3D)_[A-Z]{3}", d)
    if m is not None:
        a, b = m.group(0).split("_")
        return f"{b}{a}".lower()
    warnings.warn(f"Dataset directory {d} does not follow the lagrangebench convention.")
    return d


def _load_force_fn(path: str) -> Callable:
    """Import force.py; if jax is missing, provide a NumPy-backed stand-in for `jax.numpy`."""
    added = []
    if "jax" not in sys.modules:
        try:
            import jax  # noqa: F401
        except ImportError:
            jax_mod = types.ModuleType("jax")
            jnp = types.ModuleType("jax.numpy")
            for name in dir(np):
                if not name.startswith("_"):
                    setattr(jnp, name, getattr(np, name))
            jax_mod.numpy = jnp
            jax_mod.Array = np.ndarray
            sys.modules["jax"], sys.modules["jax.numpy"] = jax_mod, jnp
            added = ["jax", "jax.numpy"]
    try:
        spec = importlib.util.spec_from_file_location("force_module", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.force_fn
    finally:
        for k in added:
            sys.modules.pop(k, None)


def per_particle_host_force(fn: Callable):
    """vmap(external_force_fn) of features.py:105-107 for a reference-convention function
    ``fn(position (dim,)) -> (dim,)``, evaluated row by row on the host."""
    def host_fn(pos):  # (n, dim) torch tensor -> (n, dim)
        import torch
        p = pos.detach().cpu().numpy()
        return torch.from_numpy(np.stack([np.asarray(fn(r), dtype=np.float64) for r in p]))
    host_fn._lb_batched = True
    return host_fn


def force_spec_from_callable(fn: Callable, bounds, n_probe: int = 64, n_verify: int = 512):
    """Compile a single-position force function into a ForceSpec by probing it on a grid:
    constant -> ForceSpec.constant; one switch along one axis -> ForceSpec.piecewise; otherwise a
    host callable evaluated per particle (slow path)."""
    from ..engine import ForceSpec
    b = np.asarray(bounds, dtype=np.float64)
    dim = b.shape[0]
    mid = b.mean(axis=1)
    f0 = np.asarray(fn(mid), dtype=np.float64)
    const, split_axis, split_at, f_lo, f_hi = True, None, None, f0, f0
    for ax in range(dim):
        xs = np.linspace(b[ax, 0], b[ax, 1], n_probe + 2)[1:-1]
        vals = []
        for x in xs:
            p = mid.copy()
            p[ax] = x
            vals.append(np.asarray(fn(p), dtype=np.float64))
        vals = np.stack(vals)
        change = np.any(vals[1:] != vals[:-1], axis=1)
        if change.any():
            const = False
            if change.sum() != 1 or split_axis is not None:
                split_axis = -1
                break
            k = int(np.nonzero(change)[0][0])
            lo, hi = xs[k], xs[k + 1]
            for _ in range(60):  # bisect the switch position
                m = 0.5 * (lo + hi)
                p = mid.copy()
                p[ax] = m
                if np.array_equal(np.asarray(fn(p), dtype=np.float64), vals[k]):
                    lo = m
                else:
                    hi = m
            split_axis, split_at, f_lo, f_hi = ax, lo, vals[k], vals[k + 1]
    # The probes above walk axis-parallel lines through the box centre on a 64-point grid: a force band narrower than
    # 1/66 of the box, or one that depends on several coordinates, slips through.  Verify the compiled spec on random
    # points of the whole box (and on points hugging a detected switch) and fall back to the callable on ANY mismatch
    # (a silently mis-compiled force is worse than the slow path).
    def compiled(p):
        if const:
            return f0
        return f_hi if p[split_axis] > split_at else f_lo

    spec_ok = const or (split_axis is not None and split_axis >= 0)
    if spec_ok:
        rng = np.random.default_rng(0)
        pts = b[:, 0] + rng.random((n_verify, dim)) * (b[:, 1] - b[:, 0])
        if not const:
            near = pts[: n_verify // 4].copy()
            near[:, split_axis] = split_at + (rng.random(len(near)) - 0.5) * 1e-3 * (b[split_axis, 1] - b[split_axis, 0])
            pts = np.concatenate([pts, near])
        for p in pts:
            if not np.array_equal(np.asarray(fn(p), dtype=np.float64), compiled(p)):
                spec_ok = False
                break
    if spec_ok and const:
        return ForceSpec.constant(f0)
    if spec_ok:
        return ForceSpec.piecewise(split_axis, split_at, f_lo, f_hi)
    return ForceSpec.callable(per_particle_host_force(fn))


class H5Dataset:
    def __init__(self, split: str, dataset_path: str, name: Optional[str] = None, input_seq_length: int = 6,
                 extra_seq_length: int = 0, nl_backend: str = "jaxmd_vmap"):
        dataset_path = osp.normpath(dataset_path)
        self.name = get_dataset_name_from_path(dataset_path) if name is None else name
        if not osp.exists(dataset_path):
            raise FileNotFoundError(f"dataset {dataset_path} not found (automatic download is not available offline)")
        assert split in ["train", "valid", "test"]
        assert input_seq_length > 1, "To compute at least one past velocity, input_seq_length must be >= 2."
        if nl_backend == "matscipy":
            raise NotImplementedError("padded variable-N datasets (matscipy backend) are not built")
        self.dataset_path = dataset_path
        self.file_path = osp.join(dataset_path, split + ".h5")
        self.input_seq_length = input_seq_length
        self.nl_backend = nl_backend

        force_fn_path = osp.join(dataset_path, "force.py")
        if osp.exists(force_fn_path):
            self.external_force_fn = _load_force_fn(force_fn_path)
        else:
            if self.name in ["dam2d", "rpf2d", "rpf3d"]:
                raise FileNotFoundError(f"External force function not found in {dataset_path}.")
            self.external_force_fn = None
        with open(osp.join(dataset_path, "metadata.json")) as f:
            self.metadata = json.loads(f.read())
        self._db = None
        with h5.open_file(self.file_path) as f:
            self.traj_keys = sorted(f.keys())
            self.sequence_length = f[f"{self.traj_keys[0]}/position"].shape[0]

        if split == "train":
            self.subseq_length = input_seq_length + 1 + extra_seq_length
            samples_per_traj = self.sequence_length - self.subseq_length + 1
            keylens = [samples_per_traj for _ in self.traj_keys]
            self._keylen_cumulative = np.cumsum(keylens).tolist()
            self.num_samples = int(sum(keylens))
            self.getter = self.get_window
        else:
            assert extra_seq_length > 0, "extra_seq_length must be > 0 for validation and testing."
            self.subseq_length = input_seq_length + extra_seq_length
            self._split_valid_traj_into_n = self.sequence_length // self.subseq_length
            self.num_samples = self._split_valid_traj_into_n * len(self.traj_keys)
            self.getter = self.get_trajectory
        assert self.sequence_length >= self.subseq_length, (
            f"# steps in dataset trajectory ({self.sequence_length}) must be >= subsequence length "
            f"({self.subseq_length}).")

    # -- engine-side view of force.py ---------------------------------------------------
    @property
    def force_spec(self):
        if self.external_force_fn is None:
            return None
        return force_spec_from_callable(self.external_force_fn, self.metadata["bounds"])

    def _open(self):
        if self._db is None:
            self._db = h5.open_file(self.file_path)
        return self._db

    def close(self):
        """Release the HDF5 handles (the reference relies on h5py's garbage collection)."""
        if self._db is not None:
            self._db.close()
            self._db = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_trajectory(self, idx: int):
        """data.py:199-225."""
        db = self._open()
        if self._split_valid_traj_into_n > 1:
            traj_idx = idx // self._split_valid_traj_into_n
            slice_from = (idx % self._split_valid_traj_into_n) * self.subseq_length
            slice_to = slice_from + self.subseq_length
        else:
            traj_idx, slice_from, slice_to = idx, 0, self.sequence_length
        key = self.traj_keys[traj_idx]
        pos_input = np.asarray(db[f"{key}/position"][slice_from:slice_to]).transpose((1, 0, 2))
        particle_type = np.asarray(db[f"{key}/particle_type"][:])
        return pos_input, particle_type

    def get_window(self, idx: int):
        """data.py:227-257."""
        traj_idx = bisect.bisect(self._keylen_cumulative, idx)
        el_idx = idx if traj_idx == 0 else idx - self._keylen_cumulative[traj_idx - 1]
        assert el_idx >= 0
        db = self._open()
        key = self.traj_keys[traj_idx]
        pos = np.asarray(db[f"{key}/position"][el_idx:el_idx + self.subseq_length]).transpose((1, 0, 2))
        particle_type = np.asarray(db[f"{key}/particle_type"][:])
        return pos, particle_type

    def __getitem__(self, idx: int):
        return self.getter(idx)

    def __len__(self):
        return self.num_samples
