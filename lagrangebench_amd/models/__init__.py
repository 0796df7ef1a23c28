from .base import BaseModel
from .gns import GNS

__all__ = ["BaseModel", "GNS"]
