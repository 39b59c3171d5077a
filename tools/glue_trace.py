"""Which lines of valor_amd issue the torch (non-native) device kernels of a training step: a TorchDispatchMode over one eager step
(graphs off), every aten op that touches a device tensor grouped by its innermost valor_amd frame, with the operand shapes.
usage: python tools/glue_trace.py [out.txt]"""
import os
import random
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import synth  # noqa: E402
from valor_amd.engine import TrainEngine  # noqa: E402
from valor_amd.model.valor import VALOR  # noqa: E402

TASK = "pt_contra%tva%tv%ta_caption%tva%tv%ta_mlm%tva"
dev = torch.device("cuda:0")
spec = synth.base_spec()
model = VALOR({"dropout": 0.1}, spec=spec, dtype=torch.bfloat16, device=dev)
model.load_state_dict(synth.make_state_dict(spec, seed=50), strict=True)
opts = SimpleNamespace(learning_rate=1e-4, weight_decay=0.01, clip_lr=5e-7, clip_lr_text=5e-7, new_lr=0.0, decoder_lr=-1,
                       betas=[0.9, 0.98], warmup_ratio=0.1, num_train_steps=100000, scheduler="warmup_linear", grad_norm=5.0)
eng = TrainEngine(model, opts, graphs=False)
batch = synth.make_batch(spec, batch=64, frames=8, audio_slices=2, txt_len=32, seed=50)
batch["video_pixels"] = batch["video_pixels"].to(dev)
batch["audio_spectrograms"] = batch["audio_spectrograms"].to(dev)
random.seed(1)
for _ in range(3):
    eng.train_step(batch, TASK)
torch.cuda.synchronize()
import traceback  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
DEVICE_OPS = ("add", "fill", "copy", "zero", "cat", "div", "mul", "sum", "exp", "clone", "index", "gather", "where", "repeat", "sub", "neg",
              "_to_copy", "masked", "scatter", "stack", "ones", "full", "arange")
log = {}


class Tracer(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__
        ts = [a for a in list(args) + (list(out) if isinstance(out, (tuple, list)) else [out]) if isinstance(a, torch.Tensor)]
        if any(t.is_cuda for t in ts) and any(name.startswith(o) for o in DEVICE_OPS):
            frame = "(no python frame: autograd engine)"
            for f in reversed(traceback.extract_stack()):
                if "valor_amd" in f.filename or f.filename.endswith("bench.py"):
                    frame = f"{f.filename.replace(ROOT, '')}:{f.lineno} {f.name}"
                    break
            shapes = " ".join(f"{tuple(t.shape)}:{str(t.dtype).replace('torch.', '')}" for t in ts[:3] if t.is_cuda)
            k = (name, frame, shapes)
            log[k] = log.get(k, 0) + 1
        return out


with Tracer():
    eng.train_step(batch, TASK)
    torch.cuda.synchronize()
lines = [f"{'op':24s} {'calls':>5s}  frame | shapes"]
for (name, frame, shapes), c in sorted(log.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    lines.append(f"{name:24s} {c:5d}  {frame} | {shapes}")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt + "\n")
