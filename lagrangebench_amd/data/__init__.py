from .data import H5Dataset, force_spec_from_callable, get_dataset_name_from_path
from .synthetic import SyntheticDataset, make_case
from .utils import get_dataset_stats, numpy_collate

__all__ = ["H5Dataset", "SyntheticDataset", "make_case", "get_dataset_stats", "numpy_collate",
           "get_dataset_name_from_path", "force_spec_from_callable"]
