"""Build liblbhip.so (the HIP engine) in-tree for gfx950.

    python -m lagrangebench_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with
the gpurun snapshot.  -ffp-contract=off keeps the fp64 geometry bit-identical to the CPU
oracle (no fused multiply-add in the cutoff predicate / features / integrator); the MFMA
intrinsics of the network kernels are unaffected.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "liblbhip.so")
SOURCES = ["lb_api.hip", "lb_neighbor.hip", "lb_state.hip", "lb_gns.hip", "lb_edge16.hip", "lb_segnn.hip", "lb_segnn_gen.hip", "lb_segnn_msg.hip", "lb_segnn_node.hip", "lb_sinkhorn.hip", "lb_edge16v.hip", "lb_edge16w.hip", "lb_node16s.hip", "lb_gns_generic.hip", "lb_msplit.hip", "lb_train.hip"]
HEADERS = ["lb_internal.h", "lb_device.h", "lb_f16x2.h", "lb_segnn_dev.h", "lb_features.h", "lb_msplit.h", "lb_msplit_dev.h", "lb_lin32.h", "lb_train_segnn.h", os.path.join("..", "..", "include", "lbhip.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]
# per-file extras.  lb_edge16v.hip: the SLP vectoriser packs the LayerNorm / scan arithmetic into
# v_pk_*_f32 pairs, which need register-pair shuffles (v_mov) and cannot carry a DPP operand.
# lb_edge16w.hip: its GEMM loop carries ~280 micro-operations of the previous tile's epilogue as compile-time-indexed fillers;
# the fully unrolled body exceeds LLVM's default size cap for `#pragma unroll` (16 k instructions) and without the unroll the
# register arrays become scratch memory.
EXTRA_FLAGS = {"lb_edge16v.hip": ["-fno-slp-vectorize"],
               "lb_edge16w.hip": ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=10000000"], "lb_node16s.hip": ["-fno-slp-vectorize"],
               "lb_gns_generic.hip": ["-fno-slp-vectorize"], "lb_segnn_msg.hip": ["-fno-slp-vectorize"], "lb_segnn_node.hip": ["-fno-slp-vectorize"], "lb_msplit.hip": ["-fno-slp-vectorize"]}


def _rocm_root() -> str:
    """ROCm install prefix: $ROCM_PATH, else `hipconfig --rocmpath`, else /opt/rocm."""
    root = os.environ.get("ROCM_PATH")
    if root and os.path.isdir(root):
        return root
    try:
        out = subprocess.run(["hipconfig", "--rocmpath"], capture_output=True, text=True, timeout=30).stdout.strip()
        if out and os.path.isdir(out):
            return out
    except (OSError, subprocess.SubprocessError):
        pass
    return "/opt/rocm"


def _hipcc() -> str:
    for c in (os.path.join(_rocm_root(), "bin", "hipcc"), "/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc, "-c", s, "-o", o] + FLAGS + EXTRA_FLAGS.get(src, []))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        # no library besides the HIP runtime (round 5: the training step has no library GEMM left)
        libdir = os.path.join(_rocm_root(), "lib")
        run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-Wl,-rpath," + libdir])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True))
