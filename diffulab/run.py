"""Launcher for UNMODIFIED reference-style entry scripts (``import hydra`` / ``from omegaconf import ...`` / ``from diffulab...``):

    python -m diffulab.run examples/train_diffusion.py trainer.n_epoch=1
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m diffulab.run examples/train_diffusion.py

It makes ``diffulab`` importable from the repository root and, when hydra-core / omegaconf are not installed (this image has
neither, and no network), registers the small stand-ins of ``diffulab_amd.compat.hydra_shim`` under those names before the script
runs; with the real packages installed it changes nothing."""

from __future__ import annotations

import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main() -> None:
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m diffulab.run <script.py> [hydra-style overrides ...]")
    from diffulab_amd.compat import hydra_shim

    hydra_shim.install()
    script = sys.argv[1]
    sys.argv = [script] + sys.argv[2:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
