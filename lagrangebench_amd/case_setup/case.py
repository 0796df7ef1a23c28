"""Case setup: the host-side mirror of lagrangebench/case_setup/case.py.

``case_builder`` keeps the reference's signature (case.py:62-71) and returns an object with the
``CaseSetupFn`` attributes (case.py:32-59): ``allocate, preprocess, allocate_eval,
preprocess_eval, integrate, displacement, normalization_stats``.  All of them delegate to the
HIP engine (liblbhip.so); tensors are torch tensors on the engine's device.  Every function
accepts the reference's un-batched ``sample = (pos (N,T,dim), particle_type (N,))`` as well as a
batched ``(B,N,T,dim), (B,N)`` sample (the reference reaches the latter through ``vmap``).
"""
from __future__ import annotations

import warnings
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch

from ..data.utils import get_dataset_stats
from ..defaults import defaults, merge
from ..engine import ForceSpec, RolloutEngine
from .features import FeatureDict, NeighborList


def _as_tensor(x, device=None):
    if isinstance(x, torch.Tensor):
        return x
    return torch.as_tensor(np.asarray(x), device=device)


def make_displacement(box, periodic: bool):
    """displacement_fn / shift_fn of case.py:104-108 as torch functions (API parity for user code
    and MetricsComputer(dist_fn=...); the hot path uses the in-kernel versions)."""

    def _mod(x, L):
        r = torch.fmod(x, L)
        return torch.where((r != 0) & (r < 0), r + L, r)

    def displacement(a, b):
        d = a - b
        if not periodic:
            return d
        L = torch.as_tensor(np.asarray(box), dtype=d.dtype, device=d.device)
        return _mod(d + L * 0.5, L) - 0.5 * L

    def shift(r, dr):
        s = r + dr
        if not periodic:
            return s
        L = torch.as_tensor(np.asarray(box), dtype=s.dtype, device=s.device)
        return _mod(s, L)

    return displacement, shift


class CaseSetupFn:
    """Mirror of the reference dataclass (case.py:32-59), engine-backed."""

    def __init__(self, *, box, metadata, input_seq_length, cfg_neighbors, cfg_model, noise_std,
                 force: Optional[ForceSpec], dtype, device):
        self.box = np.asarray(box, dtype=np.float64)
        self.metadata = metadata
        self.input_seq_length = int(input_seq_length)
        self.cfg_neighbors = cfg_neighbors
        self.cfg_model = cfg_model
        self.dtype = dtype
        self.f32 = _is_f32(dtype)
        self.device = device
        self.force = force
        self.dim = int(len(self.box))
        self.N = int(metadata["num_particles_max"])
        self.periodic = bool(np.array(metadata["periodic_boundary_conditions"]).any())
        self.normalization_stats = get_dataset_stats(metadata, cfg_model["isotropic_norm"], noise_std,
                                                     dtype=np.float32 if self.f32 else np.float64)
        self.displacement, self.shift = make_displacement(self.box, self.periodic)
        self._engines: Dict[int, RolloutEngine] = {}

    # ---------------------------------------------------------------- engine cache
    def engine(self, batch: int) -> RolloutEngine:
        eng = self._engines.get(batch)
        if eng is None:
            s = self.normalization_stats
            md = self.metadata
            eng = RolloutEngine(
                dim=self.dim, n_particles=self.N, batch=batch, isl=self.input_seq_length,
                box=self.box, periodic=self.periodic, r_cutoff=md["default_connectivity_radius"],
                multiplier=self.cfg_neighbors["multiplier"],
                vel_mean=s["velocity"]["mean"], vel_std=s["velocity"]["std"],
                acc_mean=s["acceleration"]["mean"], acc_std=s["acceleration"]["std"],
                bounds=md["bounds"],
                has_bound=not any(md["periodic_boundary_conditions"]),
                has_vel_mag=bool(self.cfg_model["magnitude_features"]),
                force=self.force, device=self.device, geometry_f32=self.f32,
            )
            self._engines[batch] = eng
        return eng

    @staticmethod
    def _split(sample) -> Tuple[torch.Tensor, torch.Tensor, bool]:
        pos, ptype = sample
        pos, ptype = _as_tensor(pos), _as_tensor(ptype)
        batched = pos.dim() == 4
        if not batched:
            pos, ptype = pos[None], ptype[None]
        return pos, ptype, batched

    # ---------------------------------------------------------------- _preprocess (case.py:162-206)
    def _preprocess(self, sample, neighbors: Optional[NeighborList], is_allocate: bool, mode: str,
                    unroll_steps: int = 0, noise_std: float = 0.0, key=None):
        pos, ptype, batched = self._split(sample)
        isl = self.input_seq_length
        if self.f32:  # case.py:169: pos_input = jnp.asarray(sample[0], dtype=dtype)
            pos = pos.to(torch.float32)
        if mode == "train" and noise_std not in (0, 0.0) and pos.shape[2] > 1:
            # case.py:172-178: random-walk noise on the input window, targets shifted consistently
            from ..train.strats import add_gns_noise
            key, pos = add_gns_noise(key, pos.to(torch.float64), ptype, isl, float(noise_std),
                                     lambda r, dr: self.shift(r, dr))
            if self.f32:
                pos = pos.to(torch.float32)
        B = pos.shape[0]
        eng = self.engine(B)
        eng.set_particle_type(ptype)
        traj = eng.prepare_traj(pos)
        eng.load_window(traj, t0=0, step=0)
        if is_allocate:
            eng.nl_allocate()
        else:
            if neighbors is None:
                raise ValueError("preprocess needs a NeighborList (use allocate first)")
            if (eng.e_cap, eng.cell_capacity) != (neighbors.max_occupancy, neighbors.cell_capacity):
                eng.nl_set_capacity(neighbors.cell_capacity, neighbors.max_occupancy)
            eng.nl_update()
        nbrs = NeighborList(eng, batched)
        features = FeatureDict(eng, traj[:, :, :isl], batched)
        if mode == "train":
            b0 = isl - 2 + unroll_steps
            target = self._compute_target(traj[:, :, b0:b0 + 3], batched)
            return key, features, target, nbrs
        return features, nbrs

    def _compute_target(self, p3: torch.Tensor, batched: bool):
        """case.py:142-160 (training targets; plain torch - not on the inference hot path)."""
        s = self.normalization_stats
        if self.f32:  # the window holds float-representable values: the targets are float arithmetic on them
            p3 = p3.to(torch.float32)
        t = lambda a: torch.as_tensor(a, dtype=p3.dtype, device=p3.device)
        cur_v = self.displacement(p3[:, :, 1], p3[:, :, 0])
        nxt_v = self.displacement(p3[:, :, 2], p3[:, :, 1])
        acc = nxt_v - cur_v
        out = {
            "acc": (acc - t(s["acceleration"]["mean"])) / t(s["acceleration"]["std"]),
            "vel": (nxt_v - t(s["velocity"]["mean"])) / t(s["velocity"]["std"]),
            "pos": p3[:, :, -1],
        }
        return out if batched else {k: v[0] for k, v in out.items()}

    # ---------------------------------------------------------------- public API (case.py:208-228)
    def allocate(self, key, sample, noise_std: float = 0.0, unroll_steps: int = 0):
        return self._preprocess(sample, None, True, "train", unroll_steps, noise_std, key)

    def preprocess(self, key, sample, noise_std, neighbors, unroll_steps: int = 0):
        return self._preprocess(sample, neighbors, False, "train", unroll_steps, noise_std, key)

    def allocate_eval(self, sample):
        return self._preprocess(sample, None, True, "eval")

    def preprocess_eval(self, sample, neighbors):
        return self._preprocess(sample, neighbors, False, "eval")

    def integrate(self, normalized_in: Dict[str, torch.Tensor], position_sequence):
        """integrate_fn - case.py:230-259."""
        assert any(k in normalized_in for k in ["pos", "vel", "acc"])
        if "pos" in normalized_in:
            return normalized_in["pos"]
        ps = _as_tensor(position_sequence)
        if self.f32:
            ps = ps.to(torch.float32)
        batched = ps.dim() == 4
        if not batched:
            ps = ps[None]
        mode, key = (1, "vel") if "vel" in normalized_in else (0, "acc")
        pred = _as_tensor(normalized_in[key])
        if pred.dim() == 2:
            pred = pred[None]
        eng = self.engine(ps.shape[0])
        out = eng.case_integrate(mode, pred, ps)
        return out if batched else out[0]


def _force_spec(external_force_fn, bounds=None) -> Optional[ForceSpec]:
    if external_force_fn is None:
        return None
    if isinstance(external_force_fn, ForceSpec):
        return external_force_fn
    if isinstance(external_force_fn, dict):  # synthetic datasets' force description
        return ForceSpec.piecewise(external_force_fn["axis"], external_force_fn["split"],
                                   external_force_fn["f_lo"], external_force_fn["f_hi"])
    if callable(external_force_fn):
        # the reference's convention: fn(single position (dim,)) -> (dim,) (features.py:105-107 vmaps
        # it).  The published force.py files are piecewise constant: compile to a device ForceSpec.
        # Anything else is evaluated on the host per particle - never handed to the batched slot
        # as is: a function like where(r[1] > 1, ...) would index ROW 1 of an (n, dim) tensor.
        if getattr(external_force_fn, "_lb_batched", False):  # opt-in: fn((n, dim) tensor) -> (n, dim)
            return ForceSpec.callable(external_force_fn)
        if bounds is not None:
            from ..data.data import force_spec_from_callable
            try:
                return force_spec_from_callable(external_force_fn, bounds)
            except Exception as exc:  # the probe itself failed: keep the reference semantics, slowly
                warnings.warn(f"external_force_fn could not be compiled to a device ForceSpec ({exc!r}); "
                              "evaluating it per particle on the host")
        from ..data.data import per_particle_host_force
        return ForceSpec.callable(per_particle_host_force(external_force_fn))
    raise TypeError("external_force_fn must be None, a ForceSpec, a dict or a callable")


def _is_f32(dtype) -> bool:
    return str(dtype) in ("float32", "torch.float32", "<class 'numpy.float32'>") or dtype in (np.float32, torch.float32)


def case_builder(
    box,
    metadata: Dict,
    input_seq_length: int,
    cfg_neighbors=None,
    cfg_model=None,
    noise_std: float = defaults.train.noise_std,
    external_force_fn: Optional[Callable] = None,
    dtype=defaults.dtype,
    device=None,
) -> CaseSetupFn:
    """Drop-in for lagrangebench.case_setup.case_builder (case.py:62-269).

    ``external_force_fn``: a ForceSpec (device-evaluated), a dict describing a piecewise-constant
    force, or a callable ``pos (n,dim) torch tensor -> (n,dim)`` evaluated with torch each step.
    ``dtype``: "float64" (the reference default, defaults.py:22) or "float32" (what the reference's own
    tests/case_test.py runs in): in float32 mode the positions are cast to f32 on entry (case.py:169) and every
    geometry result (neighbor predicate, features, targets, integrator) is the float result bit for bit; the
    buffers and the returned tensors stay float64 holding float-representable values.
    """
    cfg_neighbors = merge(defaults.neighbors, cfg_neighbors)
    cfg_model = merge(defaults.model, cfg_model)
    if not (_is_f32(dtype) or str(dtype) in ("float64", "torch.float64", "<class 'numpy.float64'>") or dtype in (
            np.float64, torch.float64)):
        raise NotImplementedError(f"case_builder: dtype {dtype!r} (float64 and float32 are built)")
    if cfg_neighbors.multiplier < 1.25:
        warnings.warn(f"cfg_neighbors.multiplier={cfg_neighbors.multiplier} < 1.25 is very low.")
    if cfg_neighbors.backend not in ("jaxmd_vmap", "jaxmd_scan", "hip"):
        raise NotImplementedError(f"neighbor backend {cfg_neighbors.backend!r} (padded variable-N "
                                  "data, matscipy) is not built")
    return CaseSetupFn(box=box, metadata=metadata, input_seq_length=input_seq_length,
                       cfg_neighbors=cfg_neighbors, cfg_model=cfg_model, noise_std=noise_std,
                       force=_force_spec(external_force_fn, metadata.get("bounds")), dtype=dtype, device=device)
