"""Does the default valor_gemm policy (tuned on the VALOR-base shapes, K = 768 / 3072) pick the fastest kernel family on the shapes of
BASELINE configs[3] (VideoSwin-L stages C = 192 / 384 / 768 / 1536 + the BERT-large decoder, hidden 1024)? Forward (NN) and dgrad (NT) of every
linear of a block at batch 64 x 8 frames, per-call policies (lib.GemmPolicy): default | family 1 (128 x 128) | family 3 (256 x 256) |
family 4 wherever eligible; HIP events, min of 5 interleaved rounds. usage: python tools/gemm_large_shapes_ab.py [out.json]"""
import ctypes
import json
import sys

import torch

sys.path.insert(0, ".")
from valor_amd import kernels as K, lib  # noqa: E402

dev = torch.device("cuda:0")
so = lib.load()
POL = {"default": None, "fam1": lib.GemmPolicy.make(variant=1), "fam3": lib.GemmPolicy.make(variant=3), "fam4": lib.GemmPolicy.make(narrow=1)}

b = 64
LIN = []
for name, rows, C in (("swin1", b * 4 * 56 * 56, 192), ("swin2", b * 4 * 28 * 28, 384), ("swin3", b * 4 * 14 * 14, 768), ("swin4", b * 4 * 7 * 7, 1536),
                      ("dec", 8832, 1024)):
    LIN += [(f"{name}_qkv", rows, 3 * C, C), (f"{name}_proj", rows, C, C), (f"{name}_fc1", rows, 4 * C, C), (f"{name}_fc2", rows, C, 4 * C)]


def mk(shape, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * 0.05).to(torch.bfloat16).to(dev)


def time_one(fn, reps=3):
    fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


res = {}
for (name, M, N, Kd) in LIN:
    if M * max(N, Kd) * 2 >= (1 << 31):            # operands of 2 GiB and more are cut into launches by valor_gemm: time one cut
        M = ((1 << 30) // max(N, Kd)) & ~255
    x, w = mk((M, Kd), 1), mk((N, Kd), 2)
    dy = mk((M, N), 3)
    out_f, out_d = torch.empty((M, N), dtype=torch.bfloat16, device=dev), torch.empty((M, Kd), dtype=torch.bfloat16, device=dev)
    for tag, fn_of in (("fwd", lambda pol: (lambda: K.gemm(x, w, out=out_f, policy=pol))),
                       ("dgrad", lambda pol: (lambda: K.gemm(dy, w, trans_b=True, out=out_d, policy=pol)))):
        row = {}
        for rnd in range(2):
            for pn, pol in POL.items():
                t = time_one(fn_of(pol))
                row[pn] = min(row.get(pn, 1e9), t)
        fl = 2.0 * M * N * Kd
        ta, tb = (0, 0) if tag == "fwd" else (0, 1)
        mm, nn, kk = (M, N, Kd) if tag == "fwd" else (M, Kd, N)
        fam = so.valor_gemm_kernel_for(0, ta, tb, mm, nn, kk, 0)
        best = min(row, key=row.get)
        res[f"{name}_{tag}"] = {"M": mm, "N": nn, "K": kk, "default_family": fam, "us": {k: round(v, 1) for k, v in row.items()},
                                "TFLOPs_default": round(fl / row["default"] / 1e6, 1), "best": best, "default_over_best": round(row["default"] / row[best], 3)}
        print(name, tag, res[f"{name}_{tag}"], flush=True)
    del x, w, dy, out_f, out_d
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
