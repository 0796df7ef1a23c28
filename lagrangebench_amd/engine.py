"""RolloutEngine: thin Python handle over liblbhip.so (include/lbhip.h).

PyTorch is only plumbing here: tensors own the device memory that is handed to the C ABI as
raw pointers, and ``torch.cuda.current_stream()`` supplies the hipStream_t.  All arithmetic of
the hot path happens inside the HIP library.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import CaseDesc, GnsDesc, LbHipError, SegnnDesc, check, ptr


def _d3(v, fill=0.0):
    a = [fill, fill, fill]
    for i, x in enumerate(list(v)[:3]):
        a[i] = float(x)
    return _lib.D3(*a)


def _dev(t: torch.Tensor, dtype, device) -> torch.Tensor:
    return t.to(device=device, dtype=dtype).contiguous()


class ForceSpec:
    """How external_force_fn(position) (features.py:105-107) is evaluated on the device.

    * ``ForceSpec.piecewise(axis, split, f_lo, f_hi)``: f = pos[axis] > split ? f_hi : f_lo -
      covers the RPF body force (``where(r[1] > 1.0, -1, 1) * g``) and constant gravity (DAM).
    * ``ForceSpec.callable(fn)``: arbitrary ``fn(pos (n,dim) tensor) -> (n,dim)`` evaluated with
      torch on the device each step and handed to the engine as a buffer.
    """

    def __init__(self, kind: int, axis=0, split=0.0, f_lo=(0, 0, 0), f_hi=(0, 0, 0), fn=None):
        self.kind, self.axis, self.split, self.f_lo, self.f_hi, self.fn = kind, axis, split, f_lo, f_hi, fn

    @staticmethod
    def piecewise(axis: int, split: float, f_lo: Sequence[float], f_hi: Sequence[float]):
        return ForceSpec(_lib.LB_FORCE_PIECEWISE, axis, split, tuple(f_lo), tuple(f_hi))

    @staticmethod
    def constant(f: Sequence[float]):
        return ForceSpec(_lib.LB_FORCE_PIECEWISE, 0, 0.0, tuple(f), tuple(f))

    @staticmethod
    def callable(fn):
        return ForceSpec(_lib.LB_FORCE_BUFFER, fn=fn)


class RolloutEngine:
    """One engine = one (case, batch size B, device)."""

    def __init__(
        self,
        *,
        dim: int,
        n_particles: int,
        batch: int,
        isl: int,
        box: Sequence[float],
        periodic: bool,
        r_cutoff: float,
        multiplier: float,
        vel_mean, vel_std, acc_mean, acc_std,
        bounds=None,
        has_bound: bool = False,
        has_vel_mag: bool = False,
        force: Optional[ForceSpec] = None,
        device: Optional[torch.device] = None,
        geometry_f32: bool = False,
    ):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise LbHipError("RolloutEngine needs a HIP device (torch.cuda.is_available() is False); "
                             "there is no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dim, self.N, self.B, self.isl = dim, n_particles, batch, isl
        self.force = force
        d = CaseDesc()
        d.dim, d.n_particles, d.batch, d.isl = dim, n_particles, batch, isl
        d.periodic = int(bool(periodic))
        d.geometry_f32 = int(bool(geometry_f32))
        self.geometry_f32 = bool(geometry_f32)
        d.has_bound = int(bool(has_bound))
        d.has_vel_mag = int(bool(has_vel_mag))
        d.force_kind = force.kind if force is not None else _lib.LB_FORCE_NONE
        d.force_axis = force.axis if force is not None else 0
        d.box = _d3(box, 1.0)
        d.r_cutoff = float(r_cutoff)
        d.capacity_multiplier = float(multiplier)
        d.vel_mean, d.vel_std = _d3(vel_mean), _d3(vel_std, 1.0)
        d.acc_mean, d.acc_std = _d3(acc_mean), _d3(acc_std, 1.0)
        if bounds is not None:
            b = np.asarray(bounds, dtype=np.float64)
            d.bound_lo, d.bound_hi = _d3(b[:, 0]), _d3(b[:, 1])
        if force is not None and force.kind == _lib.LB_FORCE_PIECEWISE:
            d.force_split = float(force.split)
            d.force_lo, d.force_hi = _d3(force.f_lo), _d3(force.f_hi)
        self.desc = d
        self.K = isl - 1
        self.node_in = (self.K * dim + (self.K if has_vel_mag else 0) + (2 * dim if has_bound else 0)
                        + (dim if d.force_kind != _lib.LB_FORCE_NONE else 0))
        self.has_bound, self.has_vel_mag = bool(has_bound), bool(has_vel_mag)
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.current_stream(self.device)
            h = C.c_void_p()
            check(self.lib.lb_engine_create(C.byref(d), C.c_void_p(self.stream.cuda_stream), C.byref(h)),
                  "lb_engine_create")
        self._h = h
        self._keep: List[torch.Tensor] = []  # tensors the library may still read asynchronously
        self.cell_capacity = 0
        self.e_cap = 0
        self.version = 0  # bumped whenever window / list change: ties FeatureDicts to a state

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            torch.cuda.synchronize(self.device)
            self.lib.lb_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _t(self, t: Optional[torch.Tensor], dtype) -> Optional[torch.Tensor]:
        if t is None:
            return None
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t))
        t = _dev(t, dtype, self.device)
        self._keep.append(t)
        if len(self._keep) > 64:
            self._keep = self._keep[-32:]
        return t

    # ------------------------------------------------------------------ state
    def set_particle_type(self, ptype) -> None:
        t = self._t(ptype, torch.int32).reshape(self.B, self.N)
        check(self.lib.lb_set_particle_type(self._h, ptr(t)), "lb_set_particle_type")

    def prepare_traj(self, pos) -> torch.Tensor:
        """(B,N,T,dim) or (N,T,dim) positions -> fp64 contiguous device tensor."""
        t = pos if isinstance(pos, torch.Tensor) else torch.as_tensor(np.asarray(pos))
        if t.dim() == 3:
            t = t[None]
        if t.shape[0] != self.B or t.shape[1] != self.N or t.shape[3] != self.dim:
            raise ValueError(f"trajectory shape {tuple(t.shape)} does not match engine "
                             f"(B={self.B}, N={self.N}, dim={self.dim})")
        return _dev(t, torch.float64, self.device)

    def load_window(self, traj: torch.Tensor, t0: int = 0, step: int = 0) -> None:
        traj = self.prepare_traj(traj)
        self._keep.append(traj)
        check(self.lib.lb_load_window(self._h, ptr(traj), traj.shape[2], t0, step), "lb_load_window")
        self.version += 1
        self._refresh_force()

    def read_window(self) -> torch.Tensor:
        out = torch.empty((self.B, self.N, self.isl, self.dim), dtype=torch.float64, device=self.device)
        check(self.lib.lb_read_window(self._h, ptr(out)), "lb_read_window")
        return out

    def _refresh_force(self) -> None:
        if self.force is not None and self.force.kind == _lib.LB_FORCE_BUFFER:
            newest = self.read_window()[:, :, -1].reshape(self.B * self.N, self.dim)
            if self.geometry_f32:  # dtype=float32: the callable sees float32 positions and its result is a float32 array
                f = self._t(self.force.fn(newest.to(torch.float32)), torch.float32)
            else:
                f = self.force.fn(newest)
            f = self._t(f, torch.float64).reshape(self.B, self.N, self.dim)
            check(self.lib.lb_set_force(self._h, ptr(f)), "lb_set_force")

    # ------------------------------------------------------------------ neighbor list
    def nl_allocate(self) -> Tuple[int, int, List[int]]:
        cc, ec = C.c_int32(), C.c_int32()
        occ = (C.c_int32 * self.B)()
        check(self.lib.lb_nl_allocate(self._h, C.byref(cc), C.byref(ec), occ), "lb_nl_allocate")
        self.cell_capacity, self.e_cap = cc.value, ec.value
        self.version += 1
        return cc.value, ec.value, list(occ)

    def nl_set_capacity(self, cell_capacity: int, e_cap: int) -> None:
        check(self.lib.lb_nl_set_capacity(self._h, int(cell_capacity), int(e_cap)), "lb_nl_set_capacity")
        self.cell_capacity, self.e_cap = int(cell_capacity), int(e_cap)

    def nl_update(self) -> None:
        check(self.lib.lb_nl_update(self._h), "lb_nl_update")
        self.version += 1

    def nl_flags(self) -> torch.Tensor:
        out = torch.empty((self.B,), dtype=torch.int32, device=self.device)
        check(self.lib.lb_nl_read_flags(self._h, ptr(out)), "lb_nl_read_flags")
        return out

    def nl_idx(self) -> Tuple[torch.Tensor, torch.Tensor]:
        idx = torch.empty((self.B, 2, self.e_cap), dtype=torch.int32, device=self.device)
        ne = torch.empty((self.B,), dtype=torch.int32, device=self.device)
        check(self.lib.lb_nl_read_idx(self._h, ptr(idx), ptr(ne)), "lb_nl_read_idx")
        return idx, ne

    def stats(self) -> Dict[str, int]:
        n, ec, cc = C.c_int64(), C.c_int32(), C.c_int32()
        check(self.lib.lb_stats(self._h, C.byref(n), C.byref(ec), C.byref(cc)), "lb_stats")
        return {"n_edges_total": n.value, "e_cap": ec.value, "cell_capacity": cc.value}

    def edge_accounting(self, reset: bool = False) -> Dict[str, float]:
        """Edge counts of the neighbor-list builds since the last reset (include/lbhip.h: lb_edge_accounting):
        mean / first / last real E over all B trajectories, and the number of builds."""
        s, n, f, l = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        check(self.lib.lb_edge_accounting(self._h, C.byref(s), C.byref(n), C.byref(f), C.byref(l), int(bool(reset))),
              "lb_edge_accounting")
        return {"sum": s.value, "builds": n.value, "first": f.value, "last": l.value,
                "mean": s.value / n.value if n.value else float(l.value)}

    # ------------------------------------------------------------------ features

    def kernel_names(self) -> dict:
        """{"edge": ..., "node": ...}: the network kernels a GNS forward runs on at the current size."""
        buf = C.create_string_buffer(256)
        check(self.lib.lb_kernel_names(self._h, buf, 256), "lb_kernel_names")
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(";") if "=" in kv)

    def node_features(self) -> Dict[str, torch.Tensor]:
        f64 = dict(dtype=torch.float64, device=self.device)
        out = {"vel_hist": torch.empty((self.B, self.N, self.K * self.dim), **f64)}
        vm = bd = fo = None
        if self.has_vel_mag:
            vm = out["vel_mag"] = torch.empty((self.B, self.N, self.K), **f64)
        if self.has_bound:
            bd = out["bound"] = torch.empty((self.B, self.N, 2 * self.dim), **f64)
        if self.desc.force_kind != _lib.LB_FORCE_NONE:
            fo = out["force"] = torch.empty((self.B, self.N, self.dim), **f64)
        check(self.lib.lb_node_features(self._h, ptr(out["vel_hist"]), ptr(vm), ptr(bd), ptr(fo)),
              "lb_node_features")
        return out

    def edge_features(self) -> Dict[str, torch.Tensor]:
        f64 = dict(dtype=torch.float64, device=self.device)
        rd = torch.empty((self.B, self.e_cap, self.dim), **f64)
        rr = torch.empty((self.B, self.e_cap, 1), **f64)
        check(self.lib.lb_edge_features(self._h, ptr(rd), ptr(rr)), "lb_edge_features")
        return {"rel_disp": rd, "rel_dist": rr}

    # ------------------------------------------------------------------ model
    def gns_create(self, desc: GnsDesc, blob: np.ndarray) -> "GnsHandle":
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        h = C.c_void_p()
        check(self.lib.lb_gns_create(self._h, C.byref(desc), blob.ctypes.data_as(C.c_void_p),
                                     C.c_int64(blob.size), C.byref(h)), "lb_gns_create")
        return GnsHandle(self, h, desc)

    def gns_train_create(self, desc: GnsDesc, blob: np.ndarray) -> "GnsTrainHandle":
        """Device-resident training state (weights, gradients, AdamW moments) of one GNS: csrc/lb_train.hip."""
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        h = C.c_void_p()
        check(self.lib.lb_gns_train_create(self._h, C.byref(desc), blob.ctypes.data_as(C.POINTER(C.c_float)),
                                           C.c_int64(blob.size), C.byref(h)), "lb_gns_train_create")
        return GnsTrainHandle(self, h, desc, blob.size)

    def segnn_train_create(self, desc: SegnnDesc, blob: np.ndarray) -> "GnsTrainHandle":
        """Device-resident training state of one SEGNN (csrc/lb_train_segnn.h); the handle type and its zero_grad /
        loss_grad / adamw_step / read / write are the GNS ones (lb_gns_train_loss_grad dispatches on the handle)."""
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        h = C.c_void_p()
        check(self.lib.lb_segnn_train_create(self._h, C.byref(desc), blob.ctypes.data_as(C.POINTER(C.c_float)),
                                             C.c_int64(blob.size), C.byref(h)), "lb_segnn_train_create")
        return GnsTrainHandle(self, h, desc, blob.size)

    def gns_forward(self, gns: "GnsHandle", out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty((self.B, self.N, self.dim), dtype=torch.float32, device=self.device)
        check(self.lib.lb_gns_forward(self._h, gns._h, ptr(out)), "lb_gns_forward")
        return out

    def segnn_create(self, desc: SegnnDesc, blob: np.ndarray) -> "SegnnHandle":
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        h = C.c_void_p()
        check(self.lib.lb_segnn_create(self._h, C.byref(desc), blob.ctypes.data_as(C.c_void_p),
                                       C.c_int64(blob.size), C.byref(h)), "lb_segnn_create")
        return SegnnHandle(self, h, desc)

    def segnn_forward(self, segnn: "SegnnHandle", out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty((self.B, self.N, self.dim), dtype=torch.float32, device=self.device)
        check(self.lib.lb_segnn_forward(self._h, segnn._h, ptr(out)), "lb_segnn_forward")
        return out

    def math_mode(self, set_mode: int = -1) -> Tuple[int, int]:
        """(mode, guard flags): 0 exact fp32 MFMA, 1 guarded f16x2 (default), 2 unguarded f16x2; flags: 1
        large operand, 2 tiny operand tile, 4 non-finite acceleration (include/lbhip.h: lb_math_mode)."""
        mode, flags = C.c_int32(), C.c_int32()
        check(self.lib.lb_math_mode(self._h, int(set_mode), C.byref(mode), C.byref(flags)), "lb_math_mode")
        return mode.value, flags.value

    def math_fallbacks(self) -> int:
        """Steps / stand-alone forwards the range guard has redone in exact fp32 on this engine (the engine returns
        to guarded f16x2 afterwards: include/lbhip.h: lb_math_fallbacks)."""
        return int(self.lib.lb_math_fallbacks(self._h))

    def debug_inject_guard(self, flags: int, step: int) -> None:
        """Test hook: the next rollout behaves as if the range guard had raised `flags` at rollout step `step`
        (include/lbhip.h: lb_debug_inject_guard)."""
        check(self.lib.lb_debug_inject_guard(self._h, int(flags), int(step)), "lb_debug_inject_guard")

    def set_fused_aggregation(self, on: bool) -> None:
        check(self.lib.lb_set_fused_aggregation(self._h, int(bool(on))), "lb_set_fused_aggregation")

    # ------------------------------------------------------------------ integrate / rollout
    def integrate(self, acc: Optional[torch.Tensor], target: torch.Tensor,
                  pred: Optional[torch.Tensor] = None) -> None:
        acc_t = self._t(acc, torch.float32) if acc is not None else None
        tgt = self._t(target, torch.float64)
        pred_T = pred.shape[1] if pred is not None else 0
        check(self.lib.lb_integrate(self._h, ptr(acc_t), ptr(tgt), ptr(pred), pred_T), "lb_integrate")
        self.version += 1
        self._refresh_force()

    def case_integrate(self, mode: int, pred: torch.Tensor, pos_seq: torch.Tensor) -> torch.Tensor:
        p = self._t(pred, torch.float32)
        ps = self.prepare_traj(pos_seq)
        out = torch.empty((self.B, self.N, self.dim), dtype=torch.float64, device=self.device)
        check(self.lib.lb_case_integrate(self._h, mode, ptr(p), ptr(ps), ps.shape[2], ptr(out)),
              "lb_case_integrate")
        return out

    def rollout(self, model, traj: torch.Tensor, n_steps: int) -> Tuple[torch.Tensor, int]:
        """model: a GnsHandle (lb_rollout) or a SegnnHandle (lb_segnn_rollout)."""
        traj = self.prepare_traj(traj)
        pred = torch.zeros((self.B, n_steps, self.N, self.dim), dtype=torch.float64, device=self.device)
        nre = C.c_int32(0)
        fn, name = ((self.lib.lb_segnn_rollout, "lb_segnn_rollout") if isinstance(model, SegnnHandle)
                    else (self.lib.lb_rollout, "lb_rollout"))
        check(fn(self._h, model._h, ptr(traj), traj.shape[2], n_steps, ptr(pred), C.byref(nre)), name)
        self.version += 1
        st = self.stats()
        self.e_cap, self.cell_capacity = st["e_cap"], st["cell_capacity"]
        return pred, nre.value

    def metrics(self, pred: torch.Tensor, target: torch.Tensor, n_steps: int,
                want=("mse",)) -> Dict[str, torch.Tensor]:
        """pred, target: (B,T,N,dim) (or (T,N,dim) when B == 1)."""
        pred = _dev(pred if pred.dim() == 4 else pred[None], torch.float64, self.device)
        target = _dev(target if target.dim() == 4 else target[None], torch.float64, self.device)
        mse = torch.empty((self.B, n_steps), dtype=torch.float64, device=self.device) if "mse" in want else None
        mae = torch.empty((self.B, n_steps), dtype=torch.float64, device=self.device) if "mae" in want else None
        check(self.lib.lb_metrics(self._h, ptr(pred), pred.shape[1], ptr(target), target.shape[1], n_steps,
                                  ptr(mse), ptr(mae)), "lb_metrics")
        out = {}
        if mse is not None:
            out["mse"] = mse
        if mae is not None:
            out["mae"] = mae
        return out

    def ekin(self, rollout: torch.Tensor, stride: int, dt: float, dx: float) -> torch.Tensor:
        """Kinetic energy of strided frames of (B,T,N,dim) (or (T,N,dim)) positions -> (B, n_out)."""
        r = _dev(rollout if rollout.dim() == 4 else rollout[None], torch.float64, self.device)
        T = r.shape[1]
        n_out = (T - 1 + stride - 1) // stride
        out = torch.empty((self.B, n_out), dtype=torch.float64, device=self.device)
        check(self.lib.lb_ekin(self._h, ptr(r), T, int(stride), float(dt), float(dx), ptr(out), n_out), "lb_ekin")
        return out

    def sinkhorn(self, pred: torch.Tensor, target: torch.Tensor, stride: int, threshold: float = 1e-4,
                 return_iters: bool = False):
        """Sinkhorn divergence of every stride-th frame pair of (B,T,N,dim) rollouts -> (B, n_out)."""
        p = _dev(pred if pred.dim() == 4 else pred[None], torch.float64, self.device)
        t = _dev(target if target.dim() == 4 else target[None], torch.float64, self.device)
        T = min(p.shape[1], t.shape[1])
        n_out = (T + stride - 1) // stride
        out = torch.empty((self.B, n_out), dtype=torch.float64, device=self.device)
        iters = (C.c_int32 * (self.B * n_out * 3))()
        check(self.lib.lb_sinkhorn(self._h, ptr(p), p.shape[1], ptr(t), t.shape[1], int(stride), float(threshold),
                                   ptr(out), n_out, iters), "lb_sinkhorn")
        if return_iters:
            return out, np.frombuffer(iters, dtype=np.int32).reshape(self.B, n_out, 3).copy()
        return out

    def sinkhorn_pot(self, pred: torch.Tensor, target: torch.Tensor, stride: int, reg: float = 0.1,
                     num_iter_max: int = 500, stop_thr: float = 1e-5, return_info: bool = False):
        """ot_backend="pot" (metrics.py:178-196): clip(sinkhorn2_xy - 0.5 (sinkhorn2_xx + sinkhorn2_yy), 0) of every
        stride-th frame pair of (B,T,N,dim) rollouts -> (B, n_out)."""
        p = _dev(pred if pred.dim() == 4 else pred[None], torch.float64, self.device)
        t = _dev(target if target.dim() == 4 else target[None], torch.float64, self.device)
        T = min(p.shape[1], t.shape[1])
        n_out = (T + stride - 1) // stride
        out = torch.empty((self.B, n_out), dtype=torch.float64, device=self.device)
        info = (C.c_int32 * (self.B * n_out * 6))()
        check(self.lib.lb_sinkhorn_pot(self._h, ptr(p), p.shape[1], ptr(t), t.shape[1], int(stride), float(reg),
                                       int(num_iter_max), float(stop_thr), ptr(out), n_out, info), "lb_sinkhorn_pot")
        if return_info:
            return out, np.frombuffer(info, dtype=np.int32).reshape(self.B, n_out, 6).copy()
        return out

    def segment_sum(self, msg: torch.Tensor) -> torch.Tensor:
        msg = _dev(msg, torch.float32, self.device)
        out = torch.empty((self.B * self.N, msg.shape[1]), dtype=torch.float32, device=self.device)
        check(self.lib.lb_segment_sum(self._h, ptr(msg), ptr(out), msg.shape[1]), "lb_segment_sum")
        return out

    # ------------------------------------------------------------------ timers
    def timers_enable(self, on: bool = True) -> None:
        check(self.lib.lb_timers_enable(self._h, int(on)))

    def timers_reset(self) -> None:
        check(self.lib.lb_timers_reset(self._h))

    def timers(self) -> Dict[str, Tuple[float, int]]:
        out = {}
        for c in range(self.lib.lb_timer_count()):
            ms, n = C.c_double(), C.c_int64()
            check(self.lib.lb_timer_get(self._h, c, C.byref(ms), C.byref(n)))
            out[self.lib.lb_timer_name(c).decode()] = (ms.value, n.value)
        return out


class GnsHandle:
    def __init__(self, engine: RolloutEngine, h, desc: GnsDesc):
        self.engine, self._h, self.desc = engine, h, desc
        self._tap = None

    def set_tap(self, on: bool = True) -> Optional[torch.Tensor]:
        e = self.engine
        if on:
            # the kernels work on 128-wide rows (narrower latents are zero-padded): the tap buffer is
            # 128 wide, the caller sees the first latent_size columns
            self._tap = torch.zeros((self.desc.num_mp_steps + 1, e.B * e.N, 128), dtype=torch.float32,
                                    device=e.device)
            check(e.lib.lb_gns_set_tap(self._h, ptr(self._tap)))
            return self._tap[:, :, :self.desc.latent_size]
        self._tap = None
        check(e.lib.lb_gns_set_tap(self._h, None))
        return None

    def close(self):
        if self._h:
            if self.engine._h:
                torch.cuda.synchronize(self.engine.device)
            self.engine.lib.lb_gns_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GnsTrainHandle:
    """trainer.py:35-89 on the device: value_and_grad of _mse summed over the batch + optax.adamw."""

    def __init__(self, engine: RolloutEngine, h, desc: GnsDesc, n_floats: int):
        self.engine, self._h, self.desc, self.n_floats = engine, h, desc, int(n_floats)

    def loss_grad(self, target: torch.Tensor, loss_weight: float = 1.0, want_pred: bool = False):
        """target (B, N, dim) normalised accelerations -> mean per-trajectory loss (float); gradients accumulate."""
        e = self.engine
        tgt = target.to(device=e.device, dtype=torch.float32).reshape(e.B * e.N, e.dim).contiguous()
        loss = C.c_double()
        pred = torch.empty((e.B, e.N, e.dim), dtype=torch.float32, device=e.device) if want_pred else None
        check(e.lib.lb_gns_train_loss_grad(self._h, ptr(tgt), C.c_float(float(loss_weight)), C.byref(loss),
                                           ptr(pred) if want_pred else None), "lb_gns_train_loss_grad")
        return (loss.value, pred) if want_pred else loss.value

    def zero_grad(self) -> None:
        check(self.engine.lib.lb_gns_train_zero_grad(self._h), "lb_gns_train_zero_grad")

    def adamw_step(self, lr: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8, weight_decay: float = 1e-8) -> None:
        check(self.engine.lib.lb_adamw_step(self._h, C.c_float(lr), C.c_float(b1), C.c_float(b2), C.c_float(eps),
                                            C.c_float(weight_decay)), "lb_adamw_step")

    def read(self, which: str = "weights") -> np.ndarray:
        idx = {"weights": 0, "grads": 1, "m": 2, "v": 3}[which]
        out = np.empty(self.n_floats, np.float32)
        check(self.engine.lib.lb_gns_train_read(self._h, idx, out.ctypes.data_as(C.POINTER(C.c_float)),
                                                C.c_int64(out.size)), "lb_gns_train_read")
        return out

    def step_count(self) -> int:
        """AdamW steps taken on the device (optax's `count`)."""
        return int(self.engine.lib.lb_gns_train_step_count(self._h))

    def math_fallbacks(self) -> int:
        """Training steps the X range guard of the f16x2 weight-gradient kernel sent to the exact-fp32 kernels (include/lbhip.h)."""
        return int(self.engine.lib.lb_gns_train_math_fallbacks(self._h))

    def write(self, which: str, blob: np.ndarray, step: int = -1) -> None:
        idx = {"weights": 0, "grads": 1, "m": 2, "v": 3}[which]
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        check(self.engine.lib.lb_gns_train_write(self._h, idx, blob.ctypes.data_as(C.POINTER(C.c_float)),
                                                 C.c_int64(blob.size), C.c_int64(step)), "lb_gns_train_write")

    def close(self):
        if self._h:
            if self.engine._h:
                torch.cuda.synchronize(self.engine.device)
            self.engine.lib.lb_gns_train_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SegnnHandle:
    def __init__(self, engine: RolloutEngine, h, desc: SegnnDesc):
        self.engine, self._h, self.desc = engine, h, desc
        self._tap = None

    def set_tap(self, on: bool = True) -> Optional[torch.Tensor]:
        e = self.engine
        if on:
            width = int(e.lib.lb_segnn_row_floats(self._h))   # 128, or the e3nn row of the general-irreps path
            self._tap = torch.zeros((self.desc.num_mp_steps + 1, e.B * e.N, width), dtype=torch.float32,
                                    device=e.device)
            check(e.lib.lb_segnn_set_tap(self._h, ptr(self._tap)))
        else:
            self._tap = None
            check(e.lib.lb_segnn_set_tap(self._h, None))
        return self._tap

    def close(self):
        if self._h:
            if self.engine._h:
                torch.cuda.synchronize(self.engine.device)
            self.engine.lib.lb_segnn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
