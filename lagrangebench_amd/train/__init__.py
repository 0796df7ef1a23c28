"""Training: mirror of lagrangebench/train."""
from .strats import add_gns_noise, push_forward_build, push_forward_sample_steps
from .trainer import Trainer

__all__ = ["Trainer", "add_gns_noise", "push_forward_build", "push_forward_sample_steps"]
