import random, sys, os
from types import SimpleNamespace
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_graphs_gpu as T
from valor_amd import ops
dev = torch.device("cuda:0")
model, eng, batch = T._engine(dev, True, swin=True)
ops.DropoutState.reset(78); random.seed(6); np.random.seed(7)
name = "multimodal_encoder.encoder.layer.0.intermediate.dense.weight"
orig = eng.reducer.finish_backward
def fb(last=True):
    print("touched", eng.reducer.touched.get(name), "uses", (eng.reducer.uses or {}).get(name), "sunk-in-decoder",
          [sum(1 for n in c.sunk if n == name) for c in model._graph_segs.get("decoder").captured.values()] if "decoder" in model._graph_segs else None, flush=True)
    return orig(last=last)
eng.reducer.finish_backward = fb
import traceback
og = eng.reducer._on_grad
def on_grad(n):
    if n == name:
        st = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-9:-1]]
        print("   on_grad rec=", ops.GradSink.recorder is not None, st, flush=True)
    return og(n)
eng.reducer._on_grad = on_grad
ops.GradSink.listener = on_grad
osk = ops._sunk
def sunk(p_):
    if p_._arena_name == name and ops.GradSink.recorder is not None:
        st = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-9:-1]]
        print("   sunk->recorder", st, flush=True)
    return osk(p_)
ops._sunk = sunk
for step in range(5):
    try:
        eng.train_step(batch, T.TASK)
    except RuntimeError as e:
        print("step", step, "ERR", str(e)[:200]); break
