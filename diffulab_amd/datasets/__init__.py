from .base import BatchData
from .synthetic import SyntheticDataset

__all__ = ["BatchData", "SyntheticDataset"]
