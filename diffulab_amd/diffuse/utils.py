"""Carriers shared by the diffusion heads (reference diffuse/utils.py:6-28, samplers/common.py:7-32)."""

from __future__ import annotations

from typing import TypedDict

import torch
from torch import Tensor

try:
    from typing import NotRequired, Required
except ImportError:
    from typing_extensions import NotRequired, Required


class SamplingOutput(TypedDict, total=False):
    x: Required[Tensor]
    estimated_x0: NotRequired[Tensor]
    xt: NotRequired[Tensor]
    xt_mean: NotRequired[Tensor]
    xt_std: NotRequired[Tensor]
    logprob: NotRequired[Tensor]


def f32_table(table_fp64: Tensor, device: torch.device) -> Tensor:
    """fp64 schedule table -> fp32 device table: the same rounding `extract_into_tensor` applies per gather
    (diffuse/utils.py:16 `.float()`), hoisted out of the loop."""
    return table_fp64.to(torch.float32).to(device).contiguous()
