from .base import BaseModel
from .gns import GNS
from .segnn import SEGNN, node_irreps

__all__ = ["BaseModel", "GNS", "SEGNN", "node_irreps"]
