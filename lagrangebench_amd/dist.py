"""Multi-GPU: independent trajectories shard embarrassingly (the reference's only batching is a
vmap over trajectories on one device, evaluate/rollout.py:226-230).  One process per GPU; there
is NO data-path collective - torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo"
in CPU tests) is used only to gather the per-trajectory metric vectors and to agree on the
slowest rank's wall time.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def local_device(local_rank: int) -> torch.device:
    """cuda:<local_rank>; on a box with fewer GPUs than ranks (the one-GPU test box running a two-rank
    smoke test over gloo) the ranks share the devices round-robin."""
    n = torch.cuda.device_count()
    return torch.device("cuda", local_rank % n if n else 0)


def _coll_device(device: Optional[torch.device]) -> torch.device:
    """Where collective payloads live: the rank's GPU under RCCL, host memory under gloo."""
    if device is not None and dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return device if device is not None else torch.device("cpu")


def pin_rank_to_cores(local_rank: int, local_world: int) -> Optional[List[int]]:
    """One rank = one GPU = one launch thread: at B = 1 a rank issues ~90 k kernel launches per second from a single
    host thread, and 8 ranks that migrate across the host's cores (or oversubscribe it with 8 x OMP threads) lose that
    rate.  Give rank r the r-th contiguous slice of the cores this process may run on and one OpenMP thread
    (LB_DIST_PIN=0 switches it off).  Returns the cores chosen, or None."""
    if os.environ.get("LB_DIST_PIN", "1") == "0" or local_world <= 1:
        return None
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("MKL_NUM_THREADS", "1")
    try:
        torch.set_num_threads(1)
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // local_world
        if per < 1:
            return None
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        return mine
    except (AttributeError, OSError, RuntimeError):
        return None


def init(backend: Optional[str] = None) -> Tuple[int, int, int]:
    rank, local_rank, world = env_world()
    # the host driver supports dmabuf IPC only: without this RCCL fails with hipIpcGetMemHandle: invalid argument
    # (already exported on the benchmark boxes; kept here for environments that build their own)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1:
        pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    # LB_DIST_FORCE_INIT=1: initialise the process group even for one rank (smoke test of the RCCL path on a one-GPU box)
    force = os.environ.get("LB_DIST_FORCE_INIT") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            # LB_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses duplicate devices)
            backend = os.environ.get("LB_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_device(local_rank))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def group_info(device: Optional[torch.device] = None) -> dict:
    """What the LIVE process group looks like from this rank: backend, world size and the ranks an all_gather
    actually returned (bench.py records it: "did RCCL see N ranks" must be readable from the bench line)."""
    rank, _, world = env_world()
    if not dist.is_initialized():
        return {"initialized": False, "backend": None, "world_size": 1, "env_world_size": world, "ranks_seen": [0]}
    t = torch.tensor([dist.get_rank()], dtype=torch.int64, device=_coll_device(device))
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return {"initialized": True, "backend": str(dist.get_backend()), "world_size": dist.get_world_size(),
            "env_world_size": world, "ranks_seen": sorted(int(x.item()) for x in out)}


def shard_trajectories(n_trajs: int, rank: int, world: int) -> List[int]:
    """Trajectory i -> rank i % world."""
    return [i for i in range(n_trajs) if i % world == rank]


def barrier(device: Optional[torch.device] = None) -> None:
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_metrics(local: Dict[int, torch.Tensor], n_trajs: int, n_steps: int,
                   device: Optional[torch.device] = None) -> Dict[int, torch.Tensor]:
    """All-gather {trajectory index -> (n_steps,) metric vector} from every rank.  Each rank
    contributes a dense (ceil(n_trajs/world), 1 + n_steps) block [index, values...]; slots a rank
    does not own carry index -1.  Message size: 8*(1+n_steps) bytes per trajectory."""
    if not dist.is_initialized():
        return dict(local)
    world = dist.get_world_size()
    per = (n_trajs + world - 1) // world
    dev = _coll_device(device)
    block = torch.full((per, 1 + n_steps), -1.0, dtype=torch.float64, device=dev)
    for slot, (idx, v) in enumerate(sorted(local.items())):
        block[slot, 0] = float(idx)
        block[slot, 1:] = v.to(dev, torch.float64)
    out = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(out, block)
    merged: Dict[int, torch.Tensor] = {}
    for blk in out:
        for row in blk.cpu():
            if row[0] >= 0:
                merged[int(row[0])] = row[1:].clone()
    return merged


def gather_metric_dicts(local: Dict[str, dict]) -> Dict[str, dict]:
    """All-gather the per-rollout metric dictionaries of evaluate.eval_rollout
    ({"rollout_i": {"mse": (T,), "e_kin": {...}, ...}}) so that every rank returns the complete
    MetricsDict.  A few hundred bytes per trajectory: one object all_gather (RCCL / gloo), tensors
    travel as CPU tensors."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local

    def cpu(x):
        if isinstance(x, dict):
            return {k: cpu(v) for k, v in x.items()}
        return x.detach().cpu() if isinstance(x, torch.Tensor) else x
    parts: List[Optional[dict]] = [None] * dist.get_world_size()
    dist.all_gather_object(parts, cpu(local))
    merged: Dict[str, dict] = {}
    for part in parts:
        merged.update(part or {})
    return dict(sorted(merged.items(), key=lambda kv: int(kv[0].rsplit("_", 1)[1])))


def sharded_eval(run_trajs: Callable[[Sequence[int]], Dict[int, torch.Tensor]], n_trajs: int, n_steps: int,
                 device: Optional[torch.device] = None) -> Dict[int, torch.Tensor]:
    """Run ``run_trajs(my trajectory indices) -> {index: (n_steps,) metric}`` on every rank and
    return the merged dictionary on all ranks."""
    rank, _, world = env_world()
    mine = shard_trajectories(n_trajs, rank, world)
    local = run_trajs(mine) if mine else {}
    return gather_metrics(local, n_trajs, n_steps, device)
