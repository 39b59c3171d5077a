"""Throughput of caption generation (VALOR.generate_cap, model/pretrain.py:914-985) on the native decoder at the bench geometry: B clips of
8 frames + 2 audio slices, group 'tva', greedy and beam-3 decoding to max_generation_len (random weights never emit [SEP]: every row runs
the full length). Prints where the time goes: the encoders + K|V projections (once per clip) and the decoding loop (re-runs the text rows
each step like the reference with VALOR_KV_CACHE=0; two rows per sequence against the K|V cache otherwise, decode.py).
usage: [GEN_MODES=greedy,beam3] python tools/gen_bench.py out.json [batch] [max_len]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import decode, synth  # noqa: E402
from valor_amd.model.valor import VALOR  # noqa: E402

B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda:0")
spec = synth.base_spec()
model = VALOR({"dropout": 0.1}, spec=spec, dtype=torch.bfloat16, device=dev)
model.load_state_dict(synth.make_state_dict(spec, seed=50), strict=True)
batch = synth.make_batch(spec, batch=B, frames=8, audio_slices=2, txt_len=32, seed=50)
batch["video_pixels"] = batch["video_pixels"].to(dev)
batch["audio_spectrograms"] = batch["audio_spectrograms"].to(dev)
res = {"batch": B, "frames": 8, "audio_slices": 2, "max_generation_len": L, "group": "tva"}


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


with torch.no_grad():
    model.eval()
    res["encode_ms"] = round(timed(lambda: decode.encode_for_generation(model, batch, ["tva"])) * 1e3, 1)
    for name, beam in [(n, k) for n, k in (("greedy", 1), ("beam3", 3)) if n in os.environ.get("GEN_MODES", "greedy,beam3")]:
        t = timed(lambda: decode.generate_cap(model, batch, ["tva"], beam_size=beam, max_generation_len=L), reps=2)
        res[name] = {"seconds": round(t, 3), "captions_per_s": round(B / t, 1), "tokens_per_s": round(B * L / t, 1),
                     "ms_per_decoding_step": round((t * 1e3 - res["encode_ms"]) / L, 2)}
print(json.dumps(res, indent=1))
json.dump(res, open(sys.argv[1], "w"), indent=1)
