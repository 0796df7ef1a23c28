from .case import CaseSetupFn, case_builder
from .features import FeatureDict, NeighborList

__all__ = ["case_builder", "CaseSetupFn", "FeatureDict", "NeighborList"]
