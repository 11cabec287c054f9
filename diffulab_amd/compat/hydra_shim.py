"""The slice of hydra-core / omegaconf the reference's example scripts use, for images that have neither installed.

Covered (reference examples/train_diffusion.py:1-12,35-77): ``@hydra.main(version_base=None, config_path=..., config_name=...)``
with command-line overrides (``a.b=1``, ``group=option``), ``hydra.utils.instantiate`` (``_target_`` nodes, nested, extra kwargs),
``omegaconf.DictConfig`` (attribute + item access, ``.get``), ``OmegaConf.to_yaml`` / ``to_container`` / ``create``.  Composition
is ``diffulab_amd.config.load_config`` (``defaults:`` lists, ``_self_``, group overrides).  ``install()`` registers the modules
under their real names ONLY when the real packages cannot be imported."""

from __future__ import annotations

import functools
import importlib.util
import os
import sys
import types
from typing import Any, Callable

import yaml

from ..config import Config, instantiate, load_config


def _plain(node: Any) -> Any:
    if isinstance(node, dict):
        return {k: _plain(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        return [_plain(v) for v in node]
    return node


class OmegaConf:
    @staticmethod
    def to_yaml(cfg: Any, resolve: bool = False) -> str:
        return yaml.safe_dump(_plain(cfg), sort_keys=False)

    @staticmethod
    def to_container(cfg: Any, resolve: bool = False, **_: Any) -> Any:
        return _plain(cfg)

    @staticmethod
    def create(obj: Any = None) -> Any:
        from ..config import _wrap

        return _wrap(obj if obj is not None else {})


def main(version_base: Any = None, config_path: str | None = None, config_name: str | None = None) -> Callable:
    """``@hydra.main``: compose ``<dir of the decorated function's file>/<config_path>/<config_name>.yaml`` with the command-line
    overrides and call the function with the config.  Hydra's own flags are honoured: ``--config-name / -cn X`` selects another
    top-level file, ``--config-path / -cp P`` replaces the decorator's config_path (absolute, or relative to the script like the
    decorator's), ``--config-dir / -cd D`` adds a directory to the search path of the config groups (absolute or relative to the
    working directory); each also as ``--flag=value``."""

    def deco(fn: Callable) -> Callable:
        @functools.wraps(fn)
        def run(cfg: Any = None) -> Any:
            if cfg is not None:
                return fn(cfg)
            here = os.path.dirname(os.path.abspath(fn.__code__.co_filename))  # config_path is relative to the script
            name, path, extra, overrides, args = config_name, config_path, [], [], sys.argv[1:]
            flags = {"--config-name": "name", "-cn": "name", "--config-path": "path", "-cp": "path", "--config-dir": "dir", "-cd": "dir"}
            i = 0
            while i < len(args):
                a = args[i]
                flag, eq, val = a.partition("=")
                if flag in flags:
                    if not eq:
                        if i + 1 >= len(args):
                            raise ValueError(f"hydra.main: {a} needs a value")
                        val, i = args[i + 1], i + 1
                    kind = flags[flag]
                    if kind == "name":
                        name = val[:-5] if val.endswith(".yaml") else val
                    elif kind == "path":
                        path = val
                    else:
                        extra.append(os.path.abspath(val))
                elif a.startswith("~") or ("=" in a and not a.startswith("-")):
                    overrides.append(a)
                i += 1
            if name is None:
                raise ValueError("hydra.main: no config_name given")
            return fn(load_config(os.path.normpath(os.path.join(here, path or ".")), name, overrides, extra_dirs=extra))

        return run

    return deco


def _module(name: str, **attrs: Any) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__diffulab_shim__ = True
    return m


def install(force: bool = False) -> bool:
    """register ``hydra`` / ``hydra.utils`` / ``omegaconf`` stand-ins unless the real packages are importable; returns True when the
    stand-ins are (now) the registered modules"""
    have = {n: (n in sys.modules or importlib.util.find_spec(n) is not None) for n in ("hydra", "omegaconf")}
    if not force and all(have.values()):
        return bool(getattr(sys.modules.get("hydra"), "__diffulab_shim__", False))
    if force or not have["omegaconf"]:
        sys.modules["omegaconf"] = _module("omegaconf", OmegaConf=OmegaConf, DictConfig=Config, ListConfig=list)
    if force or not have["hydra"]:
        utils = _module("hydra.utils", instantiate=instantiate)
        h = _module("hydra", main=main, utils=utils)
        h.__path__ = []  # a package, so ``import hydra.utils`` works
        sys.modules["hydra"], sys.modules["hydra.utils"] = h, utils
    return True
