from .optim import FusedAdamW
from .utils import AverageMeter

__all__ = ["FusedAdamW", "AverageMeter"]
