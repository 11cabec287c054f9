"""Timestep respacing (IDDPM), host-side integer logic (reference diffuse/modelizations/utils.py:1-57)."""

from __future__ import annotations


def space_timesteps(num_timesteps: int, section_counts: str | int, ddim: bool = False) -> set[int]:
    """Indices of the original chain kept by a respaced sampler.

    `section_counts` "a,b,c" splits the chain into equal sections keeping a/b/c evenly spread steps of each;
    an int is a single section.  Reference behaviour kept bit-for-bit, including the ddim branch that only
    ever accepts stride 1 (its `raise` sits inside the loop body, utils.py:26-31).
    """
    if ddim:
        assert isinstance(section_counts, int)
        for stride in range(1, num_timesteps):
            kept = range(0, num_timesteps, stride)
            if len(kept) == section_counts:
                return set(kept)
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
    counts = [int(c) for c in section_counts.split(",")] if isinstance(section_counts, str) else [section_counts]
    per, extra = divmod(num_timesteps, len(counts))
    first, kept_all = 0, []
    for sec, want in enumerate(counts):
        length = per + (1 if sec < extra else 0)
        if length < want:
            raise ValueError(f"cannot divide section of {length} steps into {want}")
        step = 1 if want <= 1 else (length - 1) / (want - 1)
        at = 0.0
        for _ in range(want):
            kept_all.append(first + round(at))
            at += step
        first += length
    return set(kept_all)
