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
