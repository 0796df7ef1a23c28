"""Every environment switch that selects another kernel family or schedule and SURVIVED the round-6 clean-up (11 left in the
library; INTEGRATION.md section 3 lists them) must stay correct: the forward / rollout parity subset of
tests/test_gpu_parity.py is re-run in a subprocess under each GROUP of compatible switches (the switches are read once per
process).  LB_TRAIN_MATH / LB_TRAIN_SORT are covered by tests/test_train.py, LB_GUARD_MAX_FALLBACKS by the guard-loop tests."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = [
    {"LB_MATH": "f32"},                                   # exact-fp32 MFMA kernels (the range guard's fall-back)
    {"LB_FUSED_AGG": "0"},                                # stand-alone jraph.segment_sum
    {"LB_MSPLIT": "0", "LB_GUARD": "full"},               # wave-per-tile kernels also on small graphs (k_edge16v: every k-group
                                                          # of every tile range-tested)
    {"LB_MSPLIT": "0"},                                   # ... k_edge16w + k_node16s on small graphs
    {"LB_MSPLIT": "1"},                                   # M-split kernels also on large graphs
    {"LB_NL_KERNEL": "cell", "LB_GRAPH": "1"},            # workgroup-per-cell search; hipGraph replay of the step
    {"LB_NL_KERNEL": "nlc"},                              # wave-per-cell search (round 6) also for 3^2-cell stencils
    {"LB_GUARD": "sampled"},                              # rounds 2-3 guard: first tile of every wave only
    # every launch-fusion of rounds 2-4 switched off together (each unfused path is also the default at larger sizes):
    # multi-launch cell binning / neighbor build / degree scan + compaction, wave-per-receiver search, decoder and
    # node features / integrator as launches of their own
    {"LB_SMALL_FUSED": "0", "LB_NL_KERNEL": "wave"},
]

SEGNN_SWITCHES = [
    {"LB_SMALL_FUSED": "0"},                              # node prep / embedding / readout / integrator as separate launches
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
