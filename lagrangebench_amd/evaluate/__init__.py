from .metrics import MetricsComputer, averaged_metrics
from .rollout import _eval_batched_rollout, _forward_eval, eval_rollout, infer
from .utils import pkl2vtk, write_vtk

__all__ = ["MetricsComputer", "averaged_metrics", "eval_rollout", "infer", "_eval_batched_rollout",
           "_forward_eval", "pkl2vtk", "write_vtk"]
