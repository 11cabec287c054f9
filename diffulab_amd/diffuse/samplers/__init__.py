from .common import Sampler, StepResult

__all__ = ["Sampler", "StepResult"]
