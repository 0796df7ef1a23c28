"""Seeded synthetic stand-ins for the LagrangeBench datasets (there is no network: the Zenodo
files can not be fetched).  Geometry, particle counts, connectivity radii and feature layout
follow the real cases (reference data_gen/lagrangebench_data/*.sh,
notebooks/datasets.ipynb), so neighbor counts, node/edge feature widths and capacities are
representative; the dynamics are an analytic advection field, not SPH.

Every case returns an in-memory dataset object shaped like the reference's ``H5Dataset`` in
eval mode (data/data.py:33-269): ``ds[i] -> (pos (N,T,dim) float32, particle_type (N,) int32)``,
``ds.metadata`` (the keys case_builder / MetricsComputer read), ``ds.input_seq_length``,
``ds.num_samples`` and ``ds.force`` (a ForceSpec or None, standing in for ``force.py``).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np


class SyntheticDataset:
    def __init__(self, name, metadata, input_seq_length, extra_seq_length, n_trajs, make_traj,
                 force=None, force_numpy=None, multiplier=1.25, noise_std=3e-4, isotropic_norm=False):
        self.name = name
        self.metadata = metadata
        self.input_seq_length = input_seq_length
        self.extra_seq_length = extra_seq_length
        self.subseq_length = input_seq_length + extra_seq_length
        self.num_samples = n_trajs
        self._make = make_traj
        self.force = force                  # engine-side ForceSpec parameters (dict) or None
        self.external_force_fn = force_numpy  # numpy callable pos(n,dim)->(n,dim) (oracle side)
        self.multiplier = multiplier        # configs/*/base.yaml neighbors.multiplier
        self.noise_std = noise_std          # configs/*/gns.yaml train.noise_std
        self.isotropic_norm = isotropic_norm
        self._cache: Dict[int, Tuple[np.ndarray, np.ndarray]] = {}

    def __len__(self):
        return self.num_samples

    def __getitem__(self, idx: int):
        if idx < 0 or idx >= self.num_samples:
            raise IndexError(idx)
        if idx not in self._cache:
            self._cache[idx] = self._make(idx)
        return self._cache[idx]

    @property
    def box(self) -> np.ndarray:
        b = np.array(self.metadata["bounds"], dtype=np.float64)
        return b[:, 1] - b[:, 0]


def _meta(dim, n, box, pbc, rc, dx, vel_std, acc_std, T, name):
    return {
        "case": name, "solver": "synthetic", "dim": dim, "dx": dx, "dt": 1.0, "write_every": 1,
        "num_particles_max": int(n), "periodic_boundary_conditions": [bool(p) for p in pbc],
        "bounds": [[0.0, float(b)] for b in box], "default_connectivity_radius": float(rc),
        "vel_mean": [0.0] * dim, "vel_std": [float(vel_std)] * dim,
        "acc_mean": [0.0] * dim, "acc_std": [float(acc_std)] * dim,
        "sequence_length_test": int(T), "num_trajs_test": 1,
    }


def _lattice(counts, dx, origin=None):
    axes = [(np.arange(c) + 0.5) * dx for c in counts]
    g = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, len(counts))
    if origin is not None:
        g = g + np.asarray(origin)
    return g


def _advect(pos0, ptype, T, box, pbc, field, step_scale, rng, moving_vel=None, jitter=0.0):
    """x_{t+1} = wrap(x_t + step_scale * field(x_t) + jitter-noise); walls follow moving_vel."""
    N, dim = pos0.shape
    out = np.empty((N, T, dim), dtype=np.float64)
    x = pos0.copy()
    fluid = ptype == 0
    for t in range(T):
        out[:, t] = x
        dxv = step_scale * field(x)
        if jitter > 0:
            dxv = dxv + rng.normal(0.0, jitter, size=x.shape)
        dxv[~fluid] = 0.0
        if moving_vel is not None:
            dxv[ptype == 2] = moving_vel
        x = x + dxv
        if any(pbc):
            x = np.mod(x, box)
    return out.astype(np.float32)


def _tgv2d_field(L):
    k = 2 * math.pi / L
    return lambda x: np.stack([np.sin(k * x[:, 0]) * np.cos(k * x[:, 1]),
                               -np.cos(k * x[:, 0]) * np.sin(k * x[:, 1])], axis=-1)


def _tgv3d_field():
    return lambda x: np.stack([np.sin(x[:, 0]) * np.cos(x[:, 1]) * np.cos(x[:, 2]),
                               -np.cos(x[:, 0]) * np.sin(x[:, 1]) * np.cos(x[:, 2]),
                               np.zeros(len(x))], axis=-1)


def make_case(name: str, n_trajs: int = 1, extra_seq_length: int = 20, input_seq_length: int = 6,
              scale: float = 1.0, vel_amp: float = 1.0) -> SyntheticDataset:
    """name in {tgv2d, rpf2d, tgv3d, ldc3d, dam2d, small2d, small3d}.  ``scale`` multiplies the
    lattice counts per side (tests use small2d / small3d or scale < 1).  ``vel_amp`` multiplies the per-frame
    displacement of the analytic advection (and its jitter): a rollout with UNTRAINED weights is ballistic
    (x' = x + (x - x_prev) + ~0), so at the datasets' velocity scale (vel_amp 1) a 20-step rollout compresses the lattice
    and the neighbour count drifts (TGV3D: 13.6 -> 18.5 -> 17 per particle); ``vel_amp`` ~ 0.03 keeps the particles
    within half a spacing of their lattice sites over 40 steps, i.e. the neighbour count at the datasets' own value
    (13.1 per particle for TGV3D) - bench.py's `stationary` line."""
    name = name.lower()
    isl = input_seq_length
    T = isl + extra_seq_length

    def counts(c):
        return [max(4, int(round(v * scale))) for v in c]

    if name in ("tgv2d", "small2d"):
        c = counts([50, 50]) if name == "tgv2d" else [16, 16]
        dx = 1.0 / c[0]
        box = np.array([c[0] * dx, c[1] * dx])
        rc = 1.45 * dx  # 0.029 at dx = 0.02
        vel_std, acc_std = 0.318 * dx, 0.0438 * dx
        md = _meta(2, c[0] * c[1], box, [True, True], rc, dx, vel_std, acc_std, T, name)
        field = _tgv2d_field(box[0])

        def make(i):
            rng = np.random.default_rng(i)
            p0 = np.mod(_lattice(c, dx) + rng.normal(0, 0.1 * dx, size=(c[0] * c[1], 2)), box)
            pt = np.zeros(len(p0), np.int32)
            return _advect(p0, pt, T, box, [True] * 2, field, 1.4 * vel_std * vel_amp, rng, jitter=0.01 * dx * vel_amp), pt

        return SyntheticDataset(name, md, isl, extra_seq_length, n_trajs, make)

    if name == "rpf2d":
        c = counts([40, 80])
        dx = 1.0 / c[0]
        box = np.array([c[0] * dx, c[1] * dx])
        rc = 1.44 * dx  # 0.036 at dx = 0.025
        vel_std, acc_std = 0.3 * dx, 0.04 * dx
        md = _meta(2, c[0] * c[1], box, [True, True], rc, dx, vel_std, acc_std, T, name)
        half = box[1] / 2
        fmag = 1.0
        field = lambda x: np.stack([np.where(x[:, 1] > half, -1.0, 1.0) * np.sin(math.pi * (x[:, 1] % half) / half),
                                    np.zeros(len(x))], axis=-1)
        force = dict(kind="piecewise", axis=1, split=float(half), f_lo=[fmag, 0.0], f_hi=[-fmag, 0.0])
        fnp = lambda r: np.where(r[..., 1:2] > half, -1.0, 1.0) * np.array([fmag, 0.0])

        def make(i):
            rng = np.random.default_rng(i)
            p0 = np.mod(_lattice(c, dx) + rng.normal(0, 0.1 * dx, size=(c[0] * c[1], 2)), box)
            pt = np.zeros(len(p0), np.int32)
            return _advect(p0, pt, T, box, [True] * 2, field, 1.4 * vel_std * vel_amp, rng, jitter=0.01 * dx * vel_amp), pt

        return SyntheticDataset(name, md, isl, extra_seq_length, n_trajs, make, force=force, force_numpy=fnp)

    if name in ("tgv3d", "small3d"):
        c = counts([20, 20, 20]) if name == "tgv3d" else [8, 8, 8]
        L = 2 * math.pi
        dx = L / c[0]
        box = np.array([L, L, L])
        rc = 0.46 * (dx / 0.314159265)  # 0.46 at dx = 2*pi/20
        vel_std, acc_std = 0.3 * dx, 0.04 * dx
        n = c[0] * c[1] * c[2]
        md = _meta(3, n, box, [True] * 3, rc, dx, vel_std, acc_std, T, name)
        field = _tgv3d_field()

        def make(i):
            rng = np.random.default_rng(i)
            p0 = np.mod(_lattice(c, dx) + rng.normal(0, 0.1 * dx, size=(n, 3)), box)
            pt = np.zeros(n, np.int32)
            return _advect(p0, pt, T, box, [True] * 3, field, 1.4 * vel_std * vel_amp, rng, jitter=0.01 * dx * vel_amp), pt

        return SyntheticDataset(name, md, isl, extra_seq_length, n_trajs, make)

    if name == "ldc3d":
        # lid-driven cavity: fluid block 32x18x12 inside a one-particle wall shell in x,y; z periodic;
        # the top (y-max) wall row is the MOVING lid (type 2), the rest SOLID_WALL (type 1).  34 x 20 x 12 = 8160
        # particles: the dataset's count (3D_LDC_8160_10kevery100, notebooks/datasets.ipynb cell 5).
        c = counts([34, 20, 12])
        dx = 1.0 / 24.0 / max(scale, 1e-9) if scale != 1.0 else 1.0 / 24.0
        box = np.array([c[0] * dx, c[1] * dx, c[2] * dx])
        rc = 1.44 * dx  # 0.06 at dx = 1/24
        vel_std, acc_std = 0.3 * dx, 0.04 * dx
        lat = _lattice(c, dx)
        ij = np.stack(np.meshgrid(*[np.arange(v) for v in c], indexing="ij"), axis=-1).reshape(-1, 3)
        wall = (ij[:, 0] == 0) | (ij[:, 0] == c[0] - 1) | (ij[:, 1] == 0) | (ij[:, 1] == c[1] - 1)
        lid = ij[:, 1] == c[1] - 1
        ptype0 = np.where(lid, 2, np.where(wall, 1, 0)).astype(np.int32)
        n = len(lat)
        md = _meta(3, n, box, [True] * 3, rc, dx, vel_std, acc_std, T, name)
        cx, cy = box[0] / 2, box[1] / 2
        field = lambda x: np.stack([(x[:, 1] - cy) / cy, -(x[:, 0] - cx) / cx, np.zeros(len(x))], axis=-1) * \
            (np.sin(math.pi * x[:, 0] / box[0]) * np.sin(math.pi * x[:, 1] / box[1]))[:, None]

        def make(i):
            rng = np.random.default_rng(i)
            p0 = lat.copy()
            fl = ptype0 == 0
            p0[fl] += rng.normal(0, 0.08 * dx, size=(int(fl.sum()), 3))
            p0 = np.mod(p0, box)
            return _advect(p0, ptype0, T, box, [True] * 3, field, 1.2 * vel_std * vel_amp, rng,
                           moving_vel=np.array([0.09 * dx / (1.0 / 24.0), 0.0, 0.0]), jitter=0.01 * dx * vel_amp), ptype0.copy()

        return SyntheticDataset(name, md, isl, extra_seq_length, n_trajs, make, multiplier=2.0)

    if name == "dam2d":
        # dam break: fluid column 100x50 in the lower-left corner of a [5.486, 2.12] tank with a
        # wall frame; free surface => strongly non-uniform cell occupancy; gravity force feature.
        dx = 0.02 / max(scale, 1e-9) if scale != 1.0 else 0.02
        fl_c = counts([100, 50])
        box = np.array([5.486, 2.12]) * (dx / 0.02)
        rc = 1.45 * dx
        vel_std, acc_std = 0.3 * dx, 0.04 * dx
        nwx, nwy = int(box[0] / dx) - 1, int(box[1] / dx) - 1
        fluid = _lattice(fl_c, dx, origin=[2 * dx, 2 * dx])
        bottom = _lattice([nwx, 1], dx, origin=[0.5 * dx, 0.5 * dx])
        left = _lattice([1, nwy - 1], dx, origin=[0.5 * dx, 1.5 * dx])
        right = _lattice([1, nwy - 1], dx, origin=[(nwx - 0.5) * dx, 1.5 * dx])
        walls = np.concatenate([bottom, left, right])
        if scale == 1.0 and len(fluid) + len(walls) < 5740:
            # the dataset has 5740 particles (2D_DAM_5740_20kevery100: three wall layers); a second floor row below the
            # first brings the synthetic case to the same count
            extra = 5740 - len(fluid) - len(walls)
            walls = np.concatenate([walls, _lattice([extra, 1], dx, origin=[0.5 * dx, -0.5 * dx])])
        lat = np.concatenate([fluid, walls])
        ptype0 = np.concatenate([np.zeros(len(fluid), np.int32), np.ones(len(walls), np.int32)])
        n = len(lat)
        md = _meta(2, n, box, [True, True], rc, dx, vel_std, acc_std, T, name)
        g = 1.0
        field = lambda x: np.stack([np.ones(len(x)) * (x[:, 1] / (fl_c[1] * dx)), -0.15 * np.ones(len(x))], axis=-1)
        force = dict(kind="piecewise", axis=0, split=0.0, f_lo=[0.0, -g], f_hi=[0.0, -g])
        fnp = lambda r: np.zeros_like(r) + np.array([0.0, -g])

        def make(i):
            rng = np.random.default_rng(i)
            p0 = lat.copy()
            fl = ptype0 == 0
            p0[fl] += rng.normal(0, 0.08 * dx, size=(int(fl.sum()), 2))
            p0[fl, 1] = np.maximum(p0[fl, 1], 1.6 * dx)
            p0 = np.mod(p0, box)
            tr = _advect(p0, ptype0, T, box, [True, True], field, 1.0 * vel_std * vel_amp, rng, jitter=0.01 * dx * vel_amp)
            # keep the fluid above the floor (no real pressure solve here)
            tr[fl, :, 1] = np.maximum(tr[fl, :, 1], np.float32(1.3 * dx))
            return tr, ptype0.copy()

        return SyntheticDataset(name, md, isl, extra_seq_length, n_trajs, make, force=force, force_numpy=fnp,
                                multiplier=2.0, noise_std=1e-3, isotropic_norm=True)

    raise ValueError(f"unknown synthetic case {name!r}")
