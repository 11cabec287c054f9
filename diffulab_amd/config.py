"""Minimal stand-in for the Hydra pieces the reference's example scripts use (``@hydra.main`` config composition through a
``defaults:`` list and ``hydra.utils.instantiate`` of ``_target_`` nodes): hydra-core / omegaconf are not installed in this
image.  ``_target_`` paths that start with ``diffulab.`` resolve into ``diffulab_amd.`` (same class names), so the reference's
own YAML model / diffuser / trainer nodes can be used unchanged."""

from __future__ import annotations

import importlib
import re
from pathlib import Path
from typing import Any

import yaml

_SCI = re.compile(r"^[+-]?\d+(\.\d*)?[eE][+-]?\d+$")
_ALIASES = {
    "diffulab.networks.MMDiT": "diffulab_amd.networks.denoisers.MMDiT",
    "diffulab.networks.denoisers.MMDiT": "diffulab_amd.networks.denoisers.MMDiT",
    "diffulab.networks.denoisers.UNetModel": "diffulab_amd.networks.denoisers.UNetModel",
    "diffulab.networks.UNetModel": "diffulab_amd.networks.denoisers.UNetModel",
    "torch.optim.AdamW": "diffulab_amd.training.FusedAdamW",  # same update rule, one launch over the flat arena
}


class Config(dict):
    """dict with attribute access (the subset of DictConfig the examples rely on)"""

    def __getattr__(self, k: str) -> Any:
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def _wrap(v: Any) -> Any:
    if isinstance(v, dict):
        return Config({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, str) and _SCI.match(v):  # PyYAML reads "1e-4" as a string (YAML 1.1 floats need a dot)
        return float(v)
    return v


def _merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def load_config(config_dir: str | Path, name: str, overrides: list[str] | None = None) -> Config:
    """compose ``<config_dir>/<name>.yaml``: every ``- group: option`` of its ``defaults`` list loads
    ``<config_dir>/<group>/<option>.yaml`` under key ``group``; ``_self_`` marks where the file's own keys are merged;
    ``overrides`` are ``a.b.c=value`` strings (values parsed as YAML) or, like Hydra, ``group=option`` to pick another file of a
    defaults group (``optimizer=sgd``)."""
    config_dir = Path(config_dir)
    raw = yaml.safe_load((config_dir / f"{name}.yaml").read_text()) or {}
    defaults = raw.pop("defaults", [])
    raw.pop("hydra", None)
    overrides = list(overrides or [])
    for ov in list(overrides):  # group choice overrides replace the entry of the defaults list
        key, _, val = ov.partition("=")
        if "." not in key and (config_dir / key / f"{val}.yaml").exists():
            defaults = [({key: val} if isinstance(d, dict) and key in d else d) for d in defaults]
            overrides.remove(ov)
    out: dict[str, Any] = {}
    merged_self = False
    for d in defaults:
        if d == "_self_":
            _merge(out, raw)
            merged_self = True
        else:
            (group, option), = d.items()
            node = yaml.safe_load((config_dir / group / f"{option}.yaml").read_text()) or {}
            _merge(out, {group: node})
    if not merged_self:
        _merge(out, raw)
    for ov in overrides:
        key, _, val = ov.partition("=")
        cur = out
        parts = key.split(".")
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = yaml.safe_load(val)
    return _wrap(out)


def instantiate(node: dict, **kwargs: Any) -> Any:
    """``hydra.utils.instantiate``: import ``node['_target_']`` and call it with the remaining keys (+ kwargs)"""
    node = dict(node)
    target = node.pop("_target_")
    target = _ALIASES.get(target, target)
    if target.startswith("diffulab."):
        target = "diffulab_amd." + target[len("diffulab."):]
    mod, _, attr = target.rpartition(".")
    fn = getattr(importlib.import_module(mod), attr)
    args = {k: (instantiate(v) if isinstance(v, dict) and "_target_" in v else (list(v) if isinstance(v, list) else v))
            for k, v in node.items()}
    args.update(kwargs)
    return fn(**args)
