"""One optimisation step of the pretraining loop (train_utils.py::conduct_train inner body, :302-364):
forward -> sum of losses -> backward (+ overlapped gradient all-reduce) -> warmup-linear LR -> global-norm
clip -> fused AdamW. Losses stay on the device (the reference's per-step `.item()` syncs, :309, are gone;
read them when you log)."""
import gc

import torch

from . import dist as vdist
from .ops import DropoutState
from .optim import FusedAdamW, get_lr_sched


class TrainEngine:
    def __init__(self, model, opts, optimizer=None, manage_gc=True):
        self.model, self.opts = model, opts
        # The cyclic garbage collector fires in the middle of a forward pass (thousands of short-lived autograd objects per
        # step) and stalls kernel submission for 20-40 ms while the GPU drains. Collect at step boundaries instead: the
        # young generation every step, everything every 64 steps.
        self.manage_gc = manage_gc
        if manage_gc:
            gc.disable()
        self.optimizer = optimizer or FusedAdamW(model, opts)
        self.world = torch.distributed.get_world_size() if vdist.is_dist() else 1
        self.reducer = vdist.Reducer(model.arena)
        if self.world > 1:
            model.gather_fn = vdist.packed_allgather_with_grads
        self.global_step = 0
        self.grad_norm = float(getattr(opts, "grad_norm", 5.0))
        self._task = None

    def train_step(self, batch, task):
        model, opt = self.model, self.optimizer
        if task != self._task:
            self.reducer.reset_task()
            self._task = task
        model.train()
        self.reducer.prepare_backward()
        loss_dict = model(batch, task=task, compute_loss=True)
        loss = sum(loss_dict.values())
        loss.backward()
        active = self.reducer.finish_backward()
        self.global_step += 1
        if getattr(self.opts, "num_train_steps", 0):
            ratio = get_lr_sched(self.global_step, self.opts)                      # train_utils.py:344-347
            for g in opt.param_groups:
                g["lr"] = g["init_lr"] * ratio
        opt.step(active_names=active, max_grad_norm=self.grad_norm, world_size=self.world)   # clip :358-360, step :362
        loss_dict["total_loss"] = loss.detach()
        if self.manage_gc:
            gc.collect(0 if self.global_step % 64 else 2)
        return loss_dict
