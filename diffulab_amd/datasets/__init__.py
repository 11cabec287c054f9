from .base import BaseDataset, BatchData
from .cifar10 import CIFAR10Dataset
from .imagenet import ImageNetmultiAR, MultiARBatchSampler, collate_fn
from .latents import ImageNetLatentREPA
from .mnist import MNISTDataset
from .prefetch import DevicePrefetcher
from .synthetic import SyntheticDataset

__all__ = ["BaseDataset", "BatchData", "CIFAR10Dataset", "DevicePrefetcher", "ImageNetLatentREPA", "ImageNetmultiAR", "MNISTDataset",
           "MultiARBatchSampler", "SyntheticDataset", "collate_fn"]
