from .synthetic import SyntheticDataset, make_case
from .utils import get_dataset_stats, numpy_collate

__all__ = ["SyntheticDataset", "make_case", "get_dataset_stats", "numpy_collate"]
