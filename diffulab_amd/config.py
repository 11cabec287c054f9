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


class ConfigCompositionError(ValueError):
    """what Hydra raises as MissingConfigException / ConfigCompositionException: an override names a config-group option that
    does not exist, or sets a key that is not in the composed config without the ``+`` prefix"""


def _find_option(search: list[Path], group: str, option: str) -> Path | None:
    for d in search:
        f = d / group / f"{option}.yaml"
        if f.exists():
            return f
    return None


def load_config(config_dir: str | Path, name: str, overrides: list[str] | None = None,
                extra_dirs: list[str | Path] | None = None) -> Config:
    """compose ``<config_dir>/<name>.yaml``: every ``- group: option`` of its ``defaults`` list loads ``<group>/<option>.yaml`` (looked
    up in ``config_dir``, then in ``extra_dirs`` = Hydra's ``--config-dir``) under key ``group``; ``_self_`` marks where the file's own
    keys are merged.  ``overrides`` follow Hydra's grammar:
      ``group=option``   pick another file of a config group -- an option that no search directory holds RAISES (Hydra:
                         "Could not find 'group/option'"), it never degrades into a string value;
      ``a.b.c=value``    set an EXISTING key (value parsed as YAML); a key that is not in the config raises (Hydra's struct mode);
      ``+a.b.c=value``   add a new key; ``++a.b.c=value`` add or override; ``~a.b.c`` delete a key; ``+group=option`` append a group."""
    config_dir = Path(config_dir)
    search = [config_dir] + [Path(d) for d in (extra_dirs or [])]
    top = next((d / f"{name}.yaml" for d in search if (d / f"{name}.yaml").exists()), None)
    if top is None:
        raise ConfigCompositionError(f"Cannot find primary config '{name}' in {[str(d) for d in search]}")
    raw = yaml.safe_load(top.read_text()) or {}
    defaults = list(raw.pop("defaults", []))
    raw.pop("hydra", None)
    groups_in_defaults = {next(iter(d)) for d in defaults if isinstance(d, dict)}

    def is_group(key: str) -> bool:
        return "." not in key and (key in groups_in_defaults or any((d / key).is_dir() for d in search))

    value_overrides: list[tuple[str, str, str]] = []  # (prefix, dotted key, raw value)
    for ov in overrides or []:
        m = re.match(r"^(\+\+|\+|~)?([^=]+?)(?:=(.*))?$", ov, flags=re.S)
        if m is None:
            raise ConfigCompositionError(f"Error parsing override '{ov}'")
        prefix, key, val = m.group(1) or "", m.group(2), m.group(3)
        if prefix != "~" and val is None:
            raise ConfigCompositionError(f"Error parsing override '{ov}': missing '='")
        if prefix in ("", "+") and is_group(key) and not isinstance(yaml.safe_load(val), (dict, list)):
            if _find_option(search, key, val) is None:
                have = sorted({f.stem for d in search if (d / key).is_dir() for f in (d / key).glob("*.yaml")})
                raise ConfigCompositionError(f"Could not find '{key}/{val}'\n\nAvailable options in '{key}':\n\t" + "\n\t".join(have))
            if key in groups_in_defaults:
                defaults = [({key: val} if isinstance(d, dict) and key in d else d) for d in defaults]
            elif prefix == "+":
                defaults.append({key: val})
                groups_in_defaults.add(key)
            else:
                raise ConfigCompositionError(f"Could not override '{key}'.\nDid you mean to override {key} in the defaults list? "
                                             f"No match in the defaults list.  To append to your default list use +{key}={val}")
            continue
        value_overrides.append((prefix, key, val if val is not None else ""))

    out: dict[str, Any] = {}
    merged_self = False
    for d in defaults:
        if d == "_self_":
            _merge(out, raw)
            merged_self = True
        else:
            (group, option), = d.items()
            if option is None:
                continue
            f = _find_option(search, group, option)
            if f is None:
                raise ConfigCompositionError(f"In '{name}': Could not find '{group}/{option}'")
            _merge(out, {group: yaml.safe_load(f.read_text()) or {}})
    if not merged_self:
        _merge(out, raw)
    for prefix, key, val in value_overrides:
        cur = out
        parts = key.split(".")
        for i, p in enumerate(parts[:-1]):
            if not isinstance(cur.get(p), dict):
                if prefix in ("+", "++"):
                    cur[p] = {}
                else:
                    raise ConfigCompositionError(f"Could not override '{key}'.\nTo append to your config use +{key}={val}\n"
                                                 f"Key '{p}' is not in struct\n    full_key: {'.'.join(parts[: i + 1])}")
            cur = cur[p]
        leaf = parts[-1]
        if prefix == "~":
            if leaf not in cur:
                raise ConfigCompositionError(f"Could not delete from config. '{key}' does not exist.")
            del cur[leaf]
        elif prefix == "" and leaf not in cur:
            raise ConfigCompositionError(f"Could not override '{key}'.\nTo append to your config use +{key}={val}\n"
                                         f"Key '{leaf}' is not in struct\n    full_key: {key}")
        elif prefix == "+" and leaf in cur:
            raise ConfigCompositionError(f"Could not append to config. An item is already at '{key}'.")
        else:
            cur[leaf] = yaml.safe_load(val)
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
