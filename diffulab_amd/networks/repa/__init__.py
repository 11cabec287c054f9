from .perceiver_resampler import PerceiverResampler

__all__ = ["PerceiverResampler"]
