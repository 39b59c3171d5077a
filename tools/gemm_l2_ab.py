"""A/B of the 8-phase GEMM's L2 knobs at the bench shapes, in ONE process (interleaved rounds, HIP events):
   raster   valor_gemm_set_policy(4, .)  0 = row-major over all tile columns, G = groups of G tile columns, 1000 = traffic model
   store    valor_gemm_set_policy(5, .)  0 plain, 1 non-temporal output stores (the round-3 sweep also had 2 = sc1 write-through)
   nta      (round-3 sweep only: k-contiguous A operand fetched with the nt hint; the knob is gone, profiles/r03_gemm_l2_ab.json)
usage:  python tools/gemm_l2_ab.py time out.json          timing sweep (median of rounds x reps per config)
        python tools/gemm_l2_ab.py pmc  order.json         ONE launch per (shape, config) in a fixed order, for a rocprofv3 --pmc pass;
                                                           order.json lists the launches in dispatch order (tools/gemm_l2_pmc.py joins them)
"""
import itertools
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K, lib  # noqa: E402

M, W, I = 100864, 768, 3072
SHAPES = {   # name: (M, N, K, trans_b, epilogue)
    "fc1_fwd_deriv": (M, I, W, False, "deriv"),
    "fc1_fwd_plain": (M, I, W, False, "bias"),
    "qkv_fwd": (M, 3 * W, W, False, "bias"),
    "fc2_fwd": (M, W, I, False, "bias"),
    "kv_proj": (117376, 2 * W, W, False, "bias"),
    "fc2_dgrad_deriv": (M, I, W, True, "dact"),
    "fc1_dgrad": (M, W, I, True, "plain"),
}


def make(name, dev):
    Mm, N, Kd, tb, epi = SHAPES[name]
    g = torch.Generator(device="cpu").manual_seed(hash(name) % 1000)
    A = torch.randn((Mm, Kd), generator=g).to(torch.bfloat16).to(dev)
    B = (0.05 * torch.randn((Kd, N) if tb else (N, Kd), generator=g)).to(torch.bfloat16).to(dev)
    kw = dict(trans_b=tb)
    if epi in ("bias", "deriv"):
        kw["bias"] = torch.randn((N,), generator=g).to(torch.bfloat16).to(dev)
    if epi == "deriv":
        kw.update(act=lib.ACT_QUICK_GELU | lib.ACT_DERIV, want_preact=True)
    if epi == "dact":
        kw.update(act=lib.ACT_QUICK_GELU | lib.ACT_DERIV, dact_aux=torch.rand((Mm, N), generator=g).to(torch.bfloat16).to(dev))
    out = torch.empty((Mm, N), dtype=torch.bfloat16, device=dev)
    return A, B, kw, out, 2.0 * Mm * N * Kd


def run(A, B, kw, out):
    if kw.get("want_preact"):
        return K.gemm(A, B, **kw)          # allocates its two outputs (caching allocator: no device malloc after the first call)
    return K.gemm(A, B, out=out, **kw)


def main():
    mode, path = sys.argv[1], sys.argv[2]
    dev = torch.device("cuda", 0)
    so = lib.load()
    rasters = [0, 6, 1000]
    stores = [0, 1]
    ntas = [0]
    if mode == "pmc":
        names = ["fc1_fwd_plain", "qkv_fwd", "fc2_dgrad_deriv"]
        cfgs = [(r, s, 0) for r in (0, 6) for s in (0, 1)]
    else:
        names = list(SHAPES)
        cfgs = list(itertools.product(rasters, stores, ntas))
    order, res = [], {}
    for name in names:
        A, B, kw, out, flops = make(name, dev)
        run(A, B, kw, out)
        torch.cuda.synchronize()
        if mode == "pmc":
            for (r, s, n) in cfgs:
                so.valor_gemm_set_policy(4, r); so.valor_gemm_set_policy(5, s)
                run(A, B, kw, out)
                torch.cuda.synchronize()
                order.append({"shape": name, "raster": r, "store": s, "nta": n, "MNK": SHAPES[name][:3]})
            continue
        times = {c: [] for c in cfgs}
        for rnd in range(3):
            for c in cfgs:
                so.valor_gemm_set_policy(4, c[0]); so.valor_gemm_set_policy(5, c[1])
                run(A, B, kw, out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    run(A, B, kw, out)
                e1.record()
                torch.cuda.synchronize()
                times[c].append(e0.elapsed_time(e1) / 4 * 1e3)
        res[name] = {}
        base = sorted(times[(0, 0, 0)])[1]
        for c in cfgs:
            med = sorted(times[c])[1]
            res[name][f"raster{c[0]}_store{c[1]}_nta{c[2]}"] = {"us": round(med, 1), "TF": round(flops / med / 1e6, 1), "vs_base": round(base / med, 3)}
        best = max(res[name].items(), key=lambda kv: kv[1]["TF"])
        print(name, "base", res[name]["raster0_store0_nta0"], "nt stores", res[name]["raster0_store1_nta0"], "best", best, flush=True)
        del A, B, kw, out
    so.valor_gemm_set_policy(4, 0); so.valor_gemm_set_policy(5, 1000)
    json.dump(order if mode == "pmc" else res, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
