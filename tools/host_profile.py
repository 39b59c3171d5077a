"""Host-side cost of one training step: time until train_step() returns (async launches) vs. synchronized
step time, plus a cProfile of the host code. usage: python tools/host_profile.py"""
import cProfile
import pstats
import random
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
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
import os  # noqa: E402
eng = TrainEngine(model, opts, graphs=os.environ.get("VALOR_GRAPHS", "1") == "1")       # VALOR_GRAPHS=1: the encoders replay hipGraphs
batch = synth.make_batch(spec, batch=64, frames=8, audio_slices=2, txt_len=32, seed=50)
batch["video_pixels"] = batch["video_pixels"].to(dev)
batch["audio_spectrograms"] = batch["audio_spectrograms"].to(dev)
random.seed(1)
for _ in range(4):
    eng.train_step(batch, TASK)
torch.cuda.synchronize()
for i in range(3):
    t0 = time.perf_counter()
    eng.train_step(batch, TASK)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"step {i}: host returns after {1e3*(t1-t0):.1f} ms, synchronized {1e3*(t2-t0):.1f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
eng.train_step(batch, TASK)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
