"""Golden vector for the reference's gradient-accumulation behaviour, produced with HuggingFace Accelerate itself.

Runs only in the build container (accelerate 1.14.0 is installed there):

    python tests/golden/make_accum_golden.py

The reference's trainer cannot be imported (wandb / ema_pytorch / an accelerate config file are absent), but what
decides its accumulation semantics is Accelerate's gating, not its own code: ``Trainer.__init__`` builds
``Accelerator(split_batches=True, gradient_accumulation_steps=k)`` (trainers/common.py:103-109), ``train`` wraps every
``training_step`` in ``accelerator.accumulate(...)`` (base_trainer.py:308) and ``training_step`` runs, in this order,
``optimizer.zero_grad()`` -> loss -> ``accelerator.backward(loss)`` -> ``optimizer.step()`` -> ``scheduler.step()``
(base_trainer.py:138-151).  This script drives a REAL Accelerator through exactly that call order on a tiny linear model and
records the parameters after every micro-step: k = 2, two passes over a 5-batch dataloader (so the forced synchronisation on the
last batch of a pass, ``sync_with_dataloader``, is in the vector too), AdamW, a per-batch LambdaLR.

Stored: inputs, initial weights and the trajectory.  ``tests/test_trainer_host.py`` replays ``BaseTrainer.train`` on the same
data and must land on the same parameters after every micro-step (and, with DIFFULAB_TRUE_ACCUMULATION=1, must NOT).
"""

from __future__ import annotations

import os

import numpy as np
import torch
from accelerate import Accelerator

HERE = os.path.dirname(os.path.abspath(__file__))
K, N_BATCH, N_EPOCH, B, DIN, DOUT = 2, 5, 2, 4, 6, 3


def main() -> None:
    rng = np.random.default_rng(20260)
    xs = rng.standard_normal((N_BATCH, B, DIN)).astype(np.float32)
    ys = rng.standard_normal((N_BATCH, B, DOUT)).astype(np.float32)
    w0 = rng.standard_normal((DOUT, DIN)).astype(np.float32) * 0.3
    b0 = rng.standard_normal((DOUT,)).astype(np.float32) * 0.1

    model = torch.nn.Linear(DIN, DOUT)
    with torch.no_grad():
        model.weight.copy_(torch.from_numpy(w0))
        model.bias.copy_(torch.from_numpy(b0))
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0 / (1.0 + 0.5 * s))
    data = [(torch.from_numpy(xs[i]), torch.from_numpy(ys[i])) for i in range(N_BATCH)]
    loader = torch.utils.data.DataLoader(data, batch_size=None, shuffle=False)

    acc = Accelerator(split_batches=True, gradient_accumulation_steps=K, cpu=True)
    model, opt, loader, sched = acc.prepare(model, opt, loader, sched)

    traj_w, traj_b, traj_lr, traj_sync, losses = [], [], [], [], []
    for _ in range(N_EPOCH):
        for x, y in loader:
            with acc.accumulate(model):
                # ---- the statement order of the reference's training_step (base_trainer.py:138-151)
                opt.zero_grad()
                loss = torch.nn.functional.mse_loss(model(x), y)
                losses.append(loss.item())
                acc.backward(loss)
                opt.step()
                sched.step()
                # ----
            traj_sync.append(bool(acc.sync_gradients))
            traj_w.append(model.weight.detach().clone().numpy())
            traj_b.append(model.bias.detach().clone().numpy())
            traj_lr.append(opt.param_groups[0]["lr"])
    np.savez(os.path.join(HERE, "accum_k2.npz"), xs=xs, ys=ys, w0=w0, b0=b0, traj_w=np.stack(traj_w), traj_b=np.stack(traj_b),
             traj_lr=np.array(traj_lr, np.float64), traj_sync=np.array(traj_sync), losses=np.array(losses, np.float64),
             k=K, n_epoch=N_EPOCH, accelerate_version=np.array(__import__("accelerate").__version__))
    print("sync pattern:", traj_sync)
    print("lr:", traj_lr)


if __name__ == "__main__":
    main()
