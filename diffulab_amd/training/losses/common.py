"""Extra-loss plugin base (mirrors training/losses/common.py:10-25 of the reference)."""

from __future__ import annotations

from abc import ABC

import torch.nn as nn


class LossFunction(ABC, nn.Module):
    name: str = "extra_loss"

    def __init__(self) -> None:
        super().__init__()

    def set_model(self, model: nn.Module) -> None:
        """attach to the denoiser whose features the loss consumes; no-op by default"""
        pass
