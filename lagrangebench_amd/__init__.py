"""lagrangebench_amd - an MI355X-native rollout engine behind the LagrangeBench API.

Only the inference-rollout hot path is here: neighbor list, features, GNS
message passing, integrator, metrics - hand-written HIP for gfx950 in ``csrc/`` behind the
C ABI of ``include/lbhip.h``, plus this thin Python mirror of the reference's
``case_setup`` / ``models`` / ``evaluate`` interfaces.  Importing the package does not need a
GPU; creating an engine does.
"""
import os as _os

# Kernel arguments in device memory: a rollout step of a small graph is ~30 dependent launches of 5-15 us, and the
# HIP runtime's HIP_FORCE_DEV_KERNARG=0 mode measures 15 % slower there (TGV2D-2.5k: 0.435 vs 0.379 ms per step; 2 % on
# the 64 k-node batch).  1 is the default of ROCm 7.2; pinned here (effective when set before the HIP runtime starts,
# i.e. before the first `import torch`) so that an inherited environment cannot turn it off unnoticed.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import case_setup, data, evaluate, models  # noqa: F401
from .case_setup import case_builder  # noqa: F401
from .evaluate import infer  # noqa: F401
from .utils import NodeType  # noqa: F401

__version__ = "0.1.0"
