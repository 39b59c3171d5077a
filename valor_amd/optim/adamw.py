"""Fused multi-tensor AdamW over the flat arenas + LR schedule + global-norm clipping.

Replaces optim/adamw.py (HF AdamW python loop), optim/misc.py::build_optimizer (10 param groups with
per-group init_lr / weight decay), optim/sched.py (warmup_linear / warmup_cosine) and the apex-amp
fp32-master bookkeeping. Keeps the attributes conduct_train reads (train_utils.py:237-242,344-347):
param_groups[i]['init_lr'|'lr'|'weight_decay'], basic_lr, clip_lr_visual, clip_lr_text, decoder_lr,
new_lr, new_params_name; state_dict() uses the HF per-parameter layout {step, exp_avg, exp_avg_sq}.
"""
import ctypes
import math
import weakref

import torch

from .. import lib
from ..kernels import _ptr, _stream, dt_of, workspace


def warmup_cosine(x, warmup_ratio):
    """optim/sched.py:15-18"""
    if x < warmup_ratio:
        return x / warmup_ratio
    return 0.5 * (1.0 + math.cos(math.pi * x))


def warmup_linear(x, warmup_ratio):
    """optim/sched.py:27-34"""
    if x < warmup_ratio:
        return x / warmup_ratio
    return max((x - 1.0) / (warmup_ratio - 1.0), 0)


_SCHED = {"warmup_linear": warmup_linear, "warmup_cosine": warmup_cosine}


def get_lr_sched(global_step, opts):
    """optim/sched.py:37-41"""
    return _SCHED[getattr(opts, "scheduler", "warmup_linear")](global_step / opts.num_train_steps, opts.warmup_ratio)


def _get(opts, name, default):
    if isinstance(opts, dict):
        return opts.get(name, default)
    return getattr(opts, name, default)


class FusedAdamW:
    N_GROUPS = 10

    def __init__(self, model, opts):
        self.model, self.arena = model, model.arena
        lr = float(_get(opts, "learning_rate", 1e-4))
        wd = float(_get(opts, "weight_decay", 0.01))
        dec = float(_get(opts, "decoder_lr", -1))
        if dec == -1:
            dec = lr
        new_lr = float(_get(opts, "new_lr", 0.0))
        clip_lr, clip_lr_text = float(_get(opts, "clip_lr", 5e-7)), float(_get(opts, "clip_lr_text", 5e-7))
        lrs = [lr, lr, new_lr, new_lr, clip_lr, clip_lr, clip_lr_text, clip_lr_text, dec, dec]     # optim/misc.py:66-77
        self.param_groups = [{"init_lr": l, "lr": l, "weight_decay": (wd if i % 2 == 0 else 0.0), "group": i}
                             for i, l in enumerate(lrs)]
        self.betas = tuple(float(b) for b in _get(opts, "betas", (0.9, 0.98)))
        self.eps = 1e-6
        self.correct_bias = True
        self.basic_lr, self.clip_lr_visual, self.clip_lr_text = lr, clip_lr, clip_lr_text
        self.decoder_lr, self.new_lr = dec, new_lr
        self.new_params_name = list(_get(opts, "new_params_name", []) or [])
        a = self.arena
        f32 = dict(dtype=torch.float32, device=a.device)
        self.separate_master = a.dtype != torch.float32
        self.master = a.flat.float() if self.separate_master else a.flat      # fp32 parity mode: params ARE the masters
        self.exp_avg = torch.zeros(a.numel, **f32)
        self.exp_avg_sq = torch.zeros(a.numel, **f32)
        self.steps = {n: 0 for n in a.offsets}                # per-parameter Adam step (adamw.py:62-72)
        self.total_norm = torch.zeros((), **f32)
        self.gscale = torch.ones((), **f32)
        self._tables = {}
        # the model tells its optimizers when parameters are rewritten (load_state_dict / init_parameters after construction)
        if not hasattr(model, "_optimizers"):
            model._optimizers = []
        model._optimizers.append(weakref.ref(self))

    def sync_master(self):
        """Re-derive the fp32 masters from the current parameters (bf16 mode) -- element by element, and ONLY where a master no longer
        rounds to its parameter. Called by VALOR.load_state_dict / init_parameters, so weights loaded AFTER the optimizer exists cannot
        be overwritten by stale masters on the next step; masters that are still consistent with the parameters (restored by
        load_state_dict(sd with 'master') before the model's weights were re-loaded from the same checkpoint) keep their low-order
        bits, so both resume orders continue exactly like the uninterrupted run."""
        if self.separate_master:
            flat = self.arena.flat
            keep = self.master.to(flat.dtype) == flat
            torch.where(keep, self.master, flat.float(), out=self.master)

    def init_master_from(self, state_dict_fp32):
        """Seed the fp32 masters from full-precision weights (instead of the rounded bf16 parameters)."""
        if not self.separate_master:
            return
        saved = self.arena.flat
        tmp_flat = self.master
        for name, shape, refs in self.model.table:
            o, n, _ = self.arena.offsets[name]
            if len(refs) == 1 or refs[1] == "cls.decoder.weight":
                if refs[0] in state_dict_fp32:
                    tmp_flat[o:o + n].copy_(state_dict_fp32[refs[0]].reshape(-1))
            else:
                rows = n // len(refs)
                for i, r in enumerate(refs):
                    if r in state_dict_fp32:
                        tmp_flat[o + i * rows:o + (i + 1) * rows].copy_(state_dict_fp32[r].reshape(-1))
        del saved

    def zero_grad(self):
        """Gradients are zeroed by the fused update itself; an explicit call clears the whole arena."""
        self.arena.grad.zero_()

    def _table(self, active_names):
        """chunk -> group table of a set of active tensors, cached per set (multi-task mixes alternate between a handful)"""
        key = frozenset(active_names)
        t = self._tables.get(key)
        if t is None:
            if len(self._tables) >= 16:
                self._tables.pop(next(iter(self._tables)))
            t = self.arena.chunk_group_table(active=key)
            self._tables[key] = t
        return t

    def step(self, active_names=None, max_grad_norm=-1.0, world_size=1):
        """One optimizer step over the parameters that received a gradient (`active_names`; None = all).
        The arena holds gradients SUMMED over ranks: they are scaled by 1/world_size (DDP mean) and by the
        clip coefficient inside the update kernel."""
        a = self.arena
        names = list(a.offsets) if active_names is None else [n for n in a.offsets if n in active_names]
        table = self._table(names)
        dt = dt_of(a.grad)
        ws = workspace(a.device)
        lib.call("valor_grad_norm_clip", _stream(), dt, _ptr(a.grad), _ptr(table), a.numel, 1.0 / world_size,
                 float(max_grad_norm), _ptr(ws), _ptr(self.total_norm), _ptr(self.gscale))
        for n in names:
            self.steps[n] += 1
        by_step = {}
        for n in names:
            by_step.setdefault(self.steps[n], []).append(n)
        lr = (ctypes.c_float * self.N_GROUPS)(*[g["lr"] for g in self.param_groups])
        wd = (ctypes.c_float * self.N_GROUPS)(*[g["weight_decay"] for g in self.param_groups])
        # (Running the update of everything behind the video tower on the side stream, under the next step's video encoder, was measured:
        #  534.95 / 533.70 vs 534.18 / 534.45 samples/s -- no gain, the HBM-bound update and the encoder's GEMMs share the memory system;
        #  one launch on the step's stream it stays. profiles/r03_step_ab_s7_adamw_split.txt)
        for st, ns in by_step.items():
            tb = table if len(by_step) == 1 else self._table(ns)
            lib.call("valor_adamw", _stream(), dt, _ptr(self.master), _ptr(self.exp_avg), _ptr(self.exp_avg_sq), _ptr(a.grad),
                     _ptr(a.flat) if self.separate_master else None, _ptr(tb), a.numel, lr, wd, self.N_GROUPS,
                     self.betas[0], self.betas[1], self.eps, int(st), int(self.correct_bias), _ptr(self.gscale), 1)
        return self.total_norm

    # ---- the reference's optimizer checkpoint (optimizer_step_N.pt: utils/save.py:57-64 saves torch's Optimizer.state_dict() of the
    # AdamW built by optim/misc.py:13-100; train_utils.py:226-228 loads it on --resume)
    def reference_param_groups(self):
        """The 10 parameter groups of build_optimizer as lists of REFERENCE parameter names, in its order: model.named_parameters()
        order (== the reference state-dict order without aliases / buffers) filtered by the group rules of optim/misc.py:33-64."""
        from ..model.params import optimizer_group
        from ..synth import state_dict_layout
        groups = [[] for _ in range(self.N_GROUPS)]
        # build_optimizer lists its groups as basic, new, clip visual, clip text, decoder (x decay / no decay): the same
        # numbering as model.params.optimizer_group
        for key, _, kind in state_dict_layout(self.model.spec):
            if kind in ("alias", "relidx", "tied"):
                continue
            groups[optimizer_group(key, tuple(self.new_params_name))].append(key)
        return groups

    def _ref_slices(self):
        """reference parameter name -> (arena offset, numel, shape) (packed q/k/v rows split back)"""
        out = {}
        for name, shape, refs in self.model.table:
            o, n, _ = self.arena.offsets[name]
            if len(refs) == 1 or refs[1] == "cls.decoder.weight":
                out[refs[0]] = (o, n, tuple(shape), name)
            else:
                rows = shape[0] // len(refs)
                sub = (rows,) + tuple(shape[1:])
                per = n // len(refs)
                for i, r in enumerate(refs):
                    out[r] = (o + i * per, per, sub, name)
        return out

    def reference_state_dict(self):
        """torch-Optimizer-format state dict with the reference's parameter indexing, loadable by the reference's
        optimizer.load_state_dict (and by load_reference_state_dict below)."""
        groups = self.reference_param_groups()
        sl = self._ref_slices()
        state, pgs, idx = {}, [], 0
        for gi, names in enumerate(groups):
            g = self.param_groups[gi]
            pgs.append({"weight_decay": g["weight_decay"], "lr": g["lr"], "init_lr": g["init_lr"], "betas": self.betas, "eps": self.eps,
                        "correct_bias": self.correct_bias, "params": list(range(idx, idx + len(names)))})
            for r in names:
                o, n, shape, owner = sl[r]
                if self.steps[owner] > 0:
                    state[idx] = {"step": self.steps[owner], "exp_avg": self.exp_avg[o:o + n].view(shape).clone().cpu(),
                                  "exp_avg_sq": self.exp_avg_sq[o:o + n].view(shape).clone().cpu()}
                idx += 1
        return {"state": state, "param_groups": pgs}

    def load_reference_state_dict(self, sd):
        """Resume from the reference's optimizer_step_N.pt (or from reference_state_dict()). Packed q|k|v tensors take the
        step count of their parts (the reference steps them together)."""
        groups = self.reference_param_groups()
        sl = self._ref_slices()
        order = [r for names in groups for r in names]
        assert len(sd["param_groups"]) == self.N_GROUPS and sum(len(g["params"]) for g in sd["param_groups"]) == len(order), \
            "optimizer checkpoint does not match this model's parameter groups"
        for names, g in zip(groups, sd["param_groups"]):
            assert len(names) == len(g["params"])
        for idx, st in sd["state"].items():
            o, n, shape, owner = sl[order[int(idx)]]
            self.steps[owner] = int(st["step"])
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
        for g, s_ in zip(self.param_groups, sd["param_groups"]):
            g.update({k: s_[k] for k in ("init_lr", "lr", "weight_decay") if k in s_})
        self.sync_master()          # the reference's checkpoint carries no fp32 masters: re-derive them from the loaded parameters

    # ---- compact native layout (one entry per arena tensor)
    def state_dict(self):
        state = {}
        for i, (name, (o, n, shape)) in enumerate(self.arena.offsets.items()):
            if self.steps[name] > 0:
                state[i] = {"step": self.steps[name], "exp_avg": self.exp_avg[o:o + n].view(shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view(shape).clone()}
        out = {"state": state, "param_groups": [dict(g) for g in self.param_groups], "names": list(self.arena.offsets)}
        if self.separate_master:
            out["master"] = self.master.clone()         # bf16 mode: the fp32 weights the update really runs on
        return out

    def load_state_dict(self, sd):
        names = sd.get("names", list(self.arena.offsets))
        for i, st in sd["state"].items():
            name = names[int(i)]
            o, n, _ = self.arena.offsets[name]
            self.steps[name] = int(st["step"])
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
        for g, s in zip(self.param_groups, sd.get("param_groups", [])):
            g.update({k: s[k] for k in ("init_lr", "lr", "weight_decay") if k in s})
        if self.separate_master:
            if sd.get("master") is not None and sd["master"].numel() == self.master.numel():
                self.master.copy_(sd["master"])
                self.arena.flat.copy_(self.master)     # parameters = rounded masters, exactly the state that was saved
            else:
                self.sync_master()
