"""Every environment switch that selects another kernel family or schedule and SURVIVED the round-3 clean-up must stay
correct: the forward / rollout parity subset of tests/test_gpu_parity.py is re-run in a subprocess under each of them
(the switches are read once per process).  INTEGRATION.md section 3 lists them."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = [
    {"LB_MATH": "f32"},                                   # exact-fp32 MFMA kernels (the range guard's fall-back)
    {"LB_FUSED_AGG": "0"},                                # stand-alone jraph.segment_sum
    {"LB_MSPLIT": "0"},                                   # wave-per-tile kernels also on small graphs
    {"LB_MSPLIT": "1"},                                   # M-split kernels also on large graphs
    {"LB_MSPLIT": "1", "LB_MS_NODE_T": "2"},              # ... with two node tiles per iteration
    {"LB_SMALL_FUSED": "0", "LB_NL_KERNEL": "wave"},      # multi-launch cell binning, wave-per-receiver search
    {"LB_NL_KERNEL": "cell"},                             # workgroup-per-cell search
    {"LB_GRAPH": "1"},                                    # hipGraph replay of the step
    {"LB_EDGE_NT_MIN_TILES": "0"},                        # nontemporal edge-latent streams also on small graphs
    {"LB_GUARD": "full", "LB_MSPLIT": "0"},               # every tile of the wave-per-tile edge kernel range-tested
    {"LB_PERSIST": "1"},                                  # all message-passing layers in one persistent launch
    {"LB_EDGE32": "1", "LB_MSPLIT": "0"},                 # processor edge kernel on 32-edge tiles (32x32x16 MFMA)
    {"LB_NODE_T2": "2", "LB_MSPLIT": "0"},                # node kernel with two tiles per wave and weight chunk
    {"LB_NODE_Q": "2", "LB_MSPLIT": "0"},                 # node kernel with the four-slot ring of 16 KiB chunks (k_node16q)
    {"LB_CELLS_TRAJ": "0"},                               # batches: five-launch counting sort instead of one workgroup per trajectory
    {"LB_NL_ONE": "0"},                                   # four-launch neighbor build also for one small trajectory
    {"LB_NL_ONE": "0", "LB_NL_CSCAN": "0"},               # ... and degree scan / finish / compaction as separate launches
    {"LB_MS_DEC": "0"},                                   # decoder as a launch of its own also behind the M-split node kernel
    {"LB_STEP_FUSE": "0"},                                # node features / integrator as launches of their own
    {"LB_GUARD": "sampled"},                              # rounds 2-3 guard: first tile of every wave only
    {"LB_EDGE_TICKET": "1", "LB_MSPLIT": "0"},            # LDS tile tickets in the wave-per-tile edge kernel (any size)
    {"LB_EDGE_TICKET": "0", "LB_MSPLIT": "0"},            # ... and the static strided walk (any size)
    {"LB_CELLS_ONE": "0", "LB_NL_ONE": "0", "LB_NL_MID": "0"},   # multi-launch cell binning also for one mid-size trajectory
]

SEGNN_SWITCHES = [
    {"LB_SEGNN_NODE": "0"},                               # node prep / embedding / readout / integrator as separate launches
    {"LB_SEGNN_FUSED": "0"},                              # one kernel per tensor-product block + stand-alone segment_sum
]


@pytest.mark.gpu
@pytest.mark.parametrize("env", SEGNN_SWITCHES, ids=[",".join(f"{k}={v}" for k, v in e.items()) for e in SEGNN_SWITCHES])
def test_segnn_parity_subset_under_switch(env):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    sel = "(test_segnn_forward_parity and (small2d or small3d or dam2d)) or test_segnn_rollout_parity"
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_segnn.py", "-m", "gpu", "-q", "-x", "-k", sel,
                        "-p", "no:cacheprovider"], cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True,
                       timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0, f"{env}:\n{tail}"
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


@pytest.mark.gpu
@pytest.mark.parametrize("env", SWITCHES, ids=[",".join(f"{k}={v}" for k, v in e.items()) for e in SWITCHES])
def test_parity_subset_under_switch(env):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    sel = ("(test_gns_forward_parity and (small2d or small3d or dam2d)) or (test_fused_rollout_parity and small2d) "
           "or test_fused_equals_generic_loop or test_overflow_reallocation")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-q", "-x", "-k", sel,
                        "-p", "no:cacheprovider"], cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True,
                       timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-15:])
    assert r.returncode == 0, f"{env}:\n{tail}"
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
