"""world_size-2 gloo test (CPU) of the N>1 path: trajectory sharding + metric gather + max-time
all-reduce (lagrangebench_amd/dist.py).  The per-trajectory "engine" is injected - here it is the
CPU oracle acting as the checker's stand-in, the product engine needs a GPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["LB_ROOT"])
import numpy as np, torch
import torch.distributed as dist
from lagrangebench_amd import dist as lbdist
from lagrangebench_amd.data import make_case
from oracle import lb_oracle as O
from tests._common import oracle_case

rank, local_rank, world = lbdist.init(backend="gloo")
assert world == 2 and dist.is_initialized()
n_trajs, n_steps = 5, 3
ds = make_case("small2d", n_trajs=n_trajs, extra_seq_length=n_steps)
case = oracle_case(ds)
isl = ds.input_seq_length

def cheat(params, state, sample):           # zero normalised acceleration
    return {"acc": np.zeros((len(sample[1]), 2), np.float32)}, state

def run(indices):
    out = {}
    for i in indices:
        pos, pt = ds[i]
        pos = pos.astype(np.float64)
        _, nbrs = case.allocate_eval((pos[:, :isl], pt))
        _, m, _ = O.eval_batched_rollout(cheat, case, None, {}, (pos[None], pt[None]), nbrs, n_steps, isl)
        out[i] = torch.from_numpy(m[0]["mse"])
    return out

mine = lbdist.shard_trajectories(n_trajs, rank, world)
assert mine == [i for i in range(n_trajs) if i % 2 == rank]
merged = lbdist.sharded_eval(run, n_trajs, n_steps)
assert sorted(merged) == list(range(n_trajs)), sorted(merged)
ref = run(range(n_trajs))                    # every rank recomputes everything as the check
for i in range(n_trajs):
    assert torch.allclose(merged[i], ref[i].double(), rtol=0, atol=0), i
# the product's eval_rollout gathers whole per-rollout metric dictionaries (nested, e_kin style)
local = {f"rollout_{i}": {"mse": ref[i], "e_kin": {"mse": ref[i].mean()}} for i in mine}
full = lbdist.gather_metric_dicts(local)
assert list(full) == [f"rollout_{i}" for i in range(n_trajs)], list(full)
for i in range(n_trajs):
    assert torch.equal(full[f"rollout_{i}"]["mse"], ref[i]) and full[f"rollout_{i}"]["e_kin"]["mse"] == ref[i].mean()
t = lbdist.max_over_ranks(1.0 + rank)
assert t == 2.0
lbdist.barrier()
if rank == 0:
    print("DIST_OK", [float(merged[i].mean()) for i in range(n_trajs)])
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_sharding_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, LB_ROOT=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DIST_OK" in r.stdout


def test_shard_and_single_process_paths():
    import torch
    from lagrangebench_amd import dist as lbdist
    assert lbdist.shard_trajectories(8, 3, 8) == [3]
    assert lbdist.shard_trajectories(10, 1, 4) == [1, 5, 9]
    assert sum(len(lbdist.shard_trajectories(11, r, 4)) for r in range(4)) == 11
    # without an initialised process group everything degrades to the local dictionary
    local = {0: torch.ones(3), 1: torch.zeros(3)}
    assert lbdist.gather_metrics(local, 2, 3) == local
    assert lbdist.max_over_ranks(1.5) == 1.5


WORKER8 = r'''
import os, sys
sys.path.insert(0, os.environ["LB_ROOT"])
import numpy as np, torch
import torch.distributed as dist
from lagrangebench_amd import dist as lbdist
from lagrangebench_amd.data import make_case
from oracle import lb_oracle as O
from tests._common import oracle_case

rank, local_rank, world = lbdist.init(backend="gloo")
assert world == 8 and dist.is_initialized()
# per-rank pinning: every rank got its own slice of the cores (or none on a box with fewer than 8)
cores = sorted(os.sched_getaffinity(0))
allc = [None] * world
dist.all_gather_object(allc, cores)
if all(len(c) < len(allc[0]) * 8 for c in allc) and sum(len(c) for c in allc) >= 8 and len(set(map(tuple, allc))) == 8:
    flat = [c for cs in allc for c in cs]
    assert len(flat) == len(set(flat)), allc            # disjoint slices
assert torch.get_num_threads() == 1
n_steps = 2
for n_trajs in (8, 11, 5):                               # config 4's split, uneven counts, fewer trajectories than ranks
    ds = make_case("ldc3d", n_trajs=n_trajs, extra_seq_length=n_steps, scale=0.25)   # LDC3D-shaped: walls + moving lid
    case = oracle_case(ds)
    isl = ds.input_seq_length

    def cheat(params, state, sample):
        return {"acc": np.zeros((len(sample[1]), 3), np.float32)}, state

    def run(indices):
        out = {}
        for i in indices:
            pos, pt = ds[i]
            pos = pos.astype(np.float64)
            _, nbrs = case.allocate_eval((pos[:, :isl], pt))
            _, m, _ = O.eval_batched_rollout(cheat, case, None, {}, (pos[None], pt[None]), nbrs, n_steps, isl)
            out[i] = torch.from_numpy(m[0]["mse"])
        return out

    mine = lbdist.shard_trajectories(n_trajs, rank, world)
    assert mine == [i for i in range(n_trajs) if i % 8 == rank]
    merged = lbdist.sharded_eval(run, n_trajs, n_steps)
    assert sorted(merged) == list(range(n_trajs)), (n_trajs, sorted(merged))
    ref = run(mine)
    for i in mine:
        assert torch.equal(merged[i], ref[i].double()), i
    local = {f"rollout_{i}": {"mse": ref[i]} for i in mine}
    full = lbdist.gather_metric_dicts(local)
    assert list(full) == [f"rollout_{i}" for i in range(n_trajs)], list(full)
info = lbdist.group_info()
assert info["ranks_seen"] == list(range(8)) and info["world_size"] == 8, info
assert lbdist.max_over_ranks(float(rank)) == 7.0
lbdist.barrier()
if rank == 0:
    print("DIST8_OK")
dist.destroy_process_group()
'''


def test_eight_rank_gloo_ldc3d_split(tmp_path):
    """BASELINE configs[3] is '8 independent LDC3D trajectories sharded over 8 GPUs': the same split with 8 gloo
    ranks on CPU (8, 11 and 5 trajectories), per-rank core pinning and one OpenMP thread per rank."""
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8)
    env = dict(os.environ, LB_ROOT=ROOT)
    env.pop("OMP_NUM_THREADS", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DIST8_OK" in r.stdout
