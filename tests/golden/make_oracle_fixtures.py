#!/usr/bin/env python
"""Write tests/golden/oracle_fullsize/*.npz: the CPU oracle's outputs for the full-size parity tests
(tests/_fullsize_oracle.py has the what and why).  Oracle only, CPU only, runs in the container:

    python tests/golden/make_oracle_fixtures.py [name ...]

The inputs are rebuilt exactly as the tests build them (same make_case / make_params calls, same seeds); every fixture
stores a hash of them, the tests fall back to the live oracle when it does not match.  ~6 minutes on 32 cores."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from lagrangebench_amd.data import make_case  # noqa: E402
from tests import _fullsize_oracle as FO  # noqa: E402
from tests._common import make_params, make_trained_like_params  # noqa: E402


def gns_fwd(name):
    L = 10
    ds = make_case(name, n_trajs=1, extra_seq_length=20)
    pos, pt = ds[0]
    isl = ds.input_seq_length
    params = make_params(ds, num_mp_steps=L, decoder_scale=1.0)
    out = FO.gns_forward(ds, params, L)
    out["input_hash"] = np.asarray(FO._hash_inputs(pos[:, :isl], pt) + FO.params_hash(params))
    return out


def gns_roll(name):
    L, n_steps = 10, 20
    ds = make_case(name, n_trajs=1, extra_seq_length=n_steps)
    pos, pt = ds[0]
    p2 = make_params(ds, num_mp_steps=L)
    out = FO.gns_rollout(ds, p2, L, n_steps)
    out["input_hash"] = np.asarray(FO._hash_inputs(pos, pt) + FO.params_hash(p2))
    return out


def gns_pos_tgv3d():
    L, n_steps = 10, 5
    ds = make_case("tgv3d", n_trajs=1, extra_seq_length=n_steps)
    pos, pt = ds[0]
    params = make_params(ds, num_mp_steps=L)
    out = FO.gns_rollout(ds, params, L, n_steps)
    out["input_hash"] = np.asarray(FO._hash_inputs(pos, pt) + FO.params_hash(params))
    return out


def gns_roll_np_tgv2d_b2():
    L, n_steps = 10, 20
    ds = make_case("tgv2d", n_trajs=2, extra_seq_length=n_steps, scale=1.0)
    params = make_params(ds, num_mp_steps=L)
    pos2 = np.stack([ds[0][0], ds[1][0]])
    out = FO.gns_rollout(ds, params, L, n_steps, traj_ids=(0, 1), use_torch=False)
    out["input_hash"] = np.asarray(FO._hash_inputs(pos2, ds[0][1]) + FO.params_hash(params))
    return out


def gns_fwd_trained_tgv3d_b3():
    L, batch = 10, 3
    ds = make_case("tgv3d", n_trajs=batch, extra_seq_length=2)
    isl = ds.input_seq_length
    params = make_trained_like_params(ds, num_mp_steps=L, decoder_scale=1.0)
    pos = np.stack([ds[b][0] for b in range(batch)])
    pt = np.stack([ds[b][1] for b in range(batch)])
    out = FO.gns_forward(ds, params, L, traj_ids=tuple(range(batch)))
    out["input_hash"] = np.asarray(FO._hash_inputs(pos[:, :, :isl], pt) + FO.params_hash(params))
    return out


def segnn_roll_dam2d():
    from lagrangebench_amd.models import SEGNN, node_irreps
    L, n_steps = 10, 20
    ds = make_case("dam2d", n_trajs=1, extra_seq_length=n_steps)
    ds.magnitude_features = True
    isl = ds.input_seq_length
    irr = node_irreps(ds.metadata, isl, ds.external_force_fn is not None, True, False)
    model = SEGNN(irr, "1x1o+1x0e", 64, 1, 1, "1x1o", num_mp_steps=L, n_vels=isl - 1, homogeneous_particles=False)
    params = model.init_params(5)
    params["output"]["wv"] = (params["output"]["wv"] * 0.01).astype(np.float32)
    pos, pt = ds[0]
    out = FO.segnn_rollout(ds, params, n_steps, isl - 1)
    out["input_hash"] = np.asarray(FO._hash_inputs(pos, pt) + FO.params_hash(params))
    return out


JOBS = {
    "gns_fwd_rpf2d": lambda: gns_fwd("rpf2d"), "gns_fwd_tgv3d": lambda: gns_fwd("tgv3d"), "gns_fwd_ldc3d": lambda: gns_fwd("ldc3d"),
    "gns_roll_rpf2d": lambda: gns_roll("rpf2d"), "gns_roll_tgv3d": lambda: gns_roll("tgv3d"), "gns_roll_ldc3d": lambda: gns_roll("ldc3d"),
    "gns_pos_tgv3d": gns_pos_tgv3d, "gns_roll_np_tgv2d_b2": gns_roll_np_tgv2d_b2,
    "gns_fwd_trained_tgv3d_b3": gns_fwd_trained_tgv3d_b3, "segnn_roll_dam2d": segnn_roll_dam2d,
}

if __name__ == "__main__":
    import torch
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    names = sys.argv[1:] or list(JOBS)
    for n in names:
        t0 = time.time()
        FO.save(n, JOBS[n]())
        sz = os.path.getsize(os.path.join(FO.FIXDIR, n + ".npz"))
        print(f"{n}: {time.time() - t0:.1f} s, {sz / 1024:.0f} KiB", flush=True)
