"""lagrangebench_amd - an MI355X-native rollout engine behind the LagrangeBench API.

Only the inference-rollout hot path is here: neighbor list, features, GNS
message passing, integrator, metrics - hand-written HIP for gfx950 in ``csrc/`` behind the
C ABI of ``include/lbhip.h``, plus this thin Python mirror of the reference's
``case_setup`` / ``models`` / ``evaluate`` interfaces.  Importing the package does not need a
GPU; creating an engine does.
"""
from . import case_setup, data, evaluate, models  # noqa: F401
from .case_setup import case_builder  # noqa: F401
from .evaluate import infer  # noqa: F401
from .utils import NodeType  # noqa: F401

__version__ = "0.1.0"
