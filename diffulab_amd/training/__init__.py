from .ema import EMA
from .losses import LossFunction, RepaLoss
from .optim import FusedAdamW
from .trainers import BaseTrainer, Trainer
from .utils import AverageMeter

__all__ = ["FusedAdamW", "AverageMeter", "EMA", "BaseTrainer", "Trainer", "LossFunction", "RepaLoss"]
