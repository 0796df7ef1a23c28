"""The N > 1 path with the PRODUCT engine: two ranks under torchrun run `evaluate.infer` on the HIP
engine (rank-aware eval_rollout: batch b -> rank b % world, per-rollout metric dictionaries
all-gathered) and must return exactly what one process returns.  Backend: "nccl" (= RCCL) when two
GPUs are visible, else both ranks share cuda:0 and gather over "gloo" - the data path is identical,
only the metric gather changes transport.  (reference: evaluate/rollout.py:226-253 - trajectories
are independent)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["LB_ROOT"])
import numpy as np, torch
import torch.distributed as dist
from lagrangebench_amd import dist as lbdist
from lagrangebench_amd.data import make_case
from lagrangebench_amd.evaluate import infer, averaged_metrics
from lagrangebench_amd.models import GNS
from tests._common import hip_case, make_params

two_gpus = torch.cuda.device_count() >= 2
rank, local_rank, world = lbdist.init(backend="nccl" if two_gpus else "gloo")
assert world == 2 and dist.is_initialized()
dev = torch.device("cuda", local_rank if two_gpus else 0)
torch.cuda.set_device(dev)
n_trajs, n_steps, L = 5, 4, 2
ds = make_case("small3d", n_trajs=n_trajs, extra_seq_length=n_steps)
params = make_params(ds, num_mp_steps=L)
model = GNS(3, 128, 2, L, 16)
case = hip_case(ds)
cfg = {"batch_size": 2, "metrics": ["mse", "mae"], "out_type": "pkl"}
out = infer(model, case, ds, params=params, cfg_eval_infer=cfg, n_rollout_steps=n_steps,
            rollout_dir=os.environ["LB_OUT"])
assert sorted(out) == [f"rollout_{i}" for i in range(n_trajs)], sorted(out)
# the single-process answer, computed by every rank with the process group hidden from eval_rollout
import lagrangebench_amd.evaluate.rollout as R
real = torch.distributed.is_initialized
torch.distributed.is_initialized = lambda: False
try:
    ref = infer(model, case, ds, params=params, cfg_eval_infer=dict(cfg, out_type="none"), n_rollout_steps=n_steps)
finally:
    torch.distributed.is_initialized = real
for k in ref:
    for m in ("mse", "mae", "mse1"):
        assert torch.equal(out[k][m].cpu(), ref[k][m].cpu()), (k, m)
ao, ar = averaged_metrics(out), averaged_metrics(ref)
assert ao.keys() == ar.keys() and all(abs(ao[k] - ar[k]) <= 1e-12 * abs(ar[k]) + 1e-300 for k in ar), (ao, ar)
t = lbdist.max_over_ranks(1.0 + rank, dev if two_gpus else None)
assert t == 2.0
lbdist.barrier(dev)
if rank == 0:
    files = sorted(os.listdir(os.environ["LB_OUT"]))
    assert [f for f in files if f.startswith("rollout_")] == [f"rollout_{i}.pkl" for i in range(n_trajs)], files
    assert sum(f.startswith("metrics") for f in files) == 1
    print("DIST_GPU_OK", "nccl" if two_gpus else "gloo", averaged_metrics(out)["val/loss"])
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_run_infer_on_the_hip_engine(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "rollouts"
    env = dict(os.environ, LB_ROOT=ROOT, LB_OUT=str(out), OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:  # keep the workers' tracebacks (torchrun's summary hides them)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "dist_gpu_fail.log"), "w") as f:
            f.write(r.stdout + "\n=====\n" + r.stderr)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-6000:]
    assert "DIST_GPU_OK" in r.stdout


def test_bench_two_ranks_on_this_box():
    """`bench.py --gpus 2` as the driver launches it (python -m torch.distributed.run, one rank per GPU).  With
    one GPU visible the two ranks share it and the metric gather / max-over-ranks travel over gloo
    (LB_DIST_BACKEND); with two GPUs it is the RCCL path itself.  Checks the contract fields of the JSON line
    and that the aggregate counts both ranks' trajectories."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env["LB_DIST_BACKEND"] = "gloo"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "3", "--batch", "2", "--workload", "tgv2d", "--no-cpu-baseline", "--no-other-configs"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 3 and d["scaling"] == "weak"
    n = d["config"]["n_particles"]
    assert abs(d["value"] - 2 * 2 * n * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]   # whole-job aggregate
    assert d["roofline"]["frac"] > 0 and d["unit"] == "particle-steps/s"


def test_bench_one_rank_over_rccl():
    """The RCCL transport itself on this box: bench.py with a one-rank process group on backend "nccl" (= RCCL):
    library load, communicator creation, the metric all_gather and the max-over-ranks all_reduce on GPU tensors -
    everything of the N > 1 path except a second GPU."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    env = dict(os.environ, LB_DIST_FORCE_INIT="1", LB_DIST_BACKEND="nccl", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "3", "--batch", "2",
           "--workload", "tgv2d", "--no-cpu-baseline", "--no-other-configs", "--no-pmc"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and np.isfinite(d["mse20_mean"])


def test_bench_eight_ranks_on_this_box():
    """The driver's 8-GPU command on the one-GPU box: `python -m torch.distributed.run --nproc-per-node 8 bench.py
    --gpus 8` with the 8 ranks sharing the device (metric gather and max-over-ranks over gloo; on an 8-GPU node the
    same command runs over RCCL with one rank per GPU).  Contract fields, all 8 ranks seen by the live process group,
    whole-job aggregate over 8 x B trajectories (config 4: LDC3D trajectories, one per rank)."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() < 8:
        env["LB_DIST_BACKEND"] = "gloo"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3",
           "--warmup", "3", "--batch", "1", "--workload", "ldc3d", "--repeats", "2", "--no-cpu-baseline",
           "--no-other-configs"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 3 and d["warmup"] == 3 and d["scaling"] == "weak"
    assert d["dist"]["ranks_seen"] == list(range(8)) and d["dist"]["world_size"] == 8, d["dist"]
    n = d["config"]["n_particles"]
    assert n == 8160
    assert abs(d["value"] - 8 * 1 * n * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]   # whole-job aggregate
    assert np.isfinite(d["mse20_mean"])
