from .base_trainer import BaseTrainer
from .common import Trainer

__all__ = ["BaseTrainer", "Trainer"]
