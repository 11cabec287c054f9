"""Import-name compatibility with the reference: ``import diffulab`` IS ``diffulab_amd``.

``from diffulab.diffuse import Diffuser``, ``from diffulab.training import BaseTrainer`` (examples/train_diffusion.py:7-8 of the
reference) and Hydra ``_target_`` strings such as ``diffulab.networks.MMDiT`` / ``diffulab.datasets.MNISTDataset`` resolve to the
MI355X build without touching the calling script.  Every ``diffulab.x.y`` name is bound to the SAME module object as
``diffulab_amd.x.y`` (no second copy of any module is ever executed): a meta-path finder intercepts the dotted names and hands
back the already-imported ``diffulab_amd`` module.
"""

from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import sys

import diffulab_amd as _real

_PREFIX, _REAL = __name__ + ".", _real.__name__ + "."


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real_name: str) -> None:
        self.real_name = real_name

    def create_module(self, spec):
        return importlib.import_module(self.real_name)

    def exec_module(self, module) -> None:  # already executed under its real name
        return None


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real_name = _REAL + fullname[len(_PREFIX):]
        try:
            real = importlib.import_module(real_name)
        except ModuleNotFoundError as e:
            if e.name and real_name.startswith(e.name):
                return None  # no such sub-module in the MI355X build: a normal ImportError follows
            raise
        spec = importlib.machinery.ModuleSpec(fullname, _AliasLoader(real_name), is_package=hasattr(real, "__path__"))
        return spec


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())


def __getattr__(name: str):  # diffulab.Diffuser, diffulab.MMDiT, diffulab.datasets, ...
    try:
        return getattr(_real, name)
    except AttributeError:
        return importlib.import_module(_PREFIX + name)


def __dir__():
    return sorted(set(dir(_real)))
