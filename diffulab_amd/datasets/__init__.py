from .base import BaseDataset, BatchData
from .cifar10 import CIFAR10Dataset
from .latents import ImageNetLatentREPA
from .mnist import MNISTDataset
from .prefetch import DevicePrefetcher
from .synthetic import SyntheticDataset

__all__ = ["BaseDataset", "BatchData", "CIFAR10Dataset", "DevicePrefetcher", "ImageNetLatentREPA", "MNISTDataset", "SyntheticDataset"]
