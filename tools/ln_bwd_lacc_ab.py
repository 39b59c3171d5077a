"""A/B of the fused LayerNorm BACKWARD at 768 columns: variant 1 = one wave per row, 8-byte accesses, column accumulators in registers
(the default), variant 2 = half a wave per row, 16-byte accesses, accumulators in registers (194 VGPRs: two waves per SIMD), variant 3 =
the same with the three column accumulators in a per-wave LDS array (120 VGPRs: four waves per SIMD; csrc/layernorm.hip LACC).
Outputs are compared first (dx / dres bit-identical between the half-wave variants; column sums to fp32 rounding). Interleaved rounds,
HIP events. usage: python tools/ln_bwd_lacc_ab.py [out.json]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from valor_amd import kernels as K, lib  # noqa: E402

dev = torch.device("cuda:0")
so = lib.load()
res_all = {}
import os
CASES = [(100864, 0.0, True), (100864, 0.1, True), (100864, 0.0, False), (16512, 0.1, True), (8832, 0.1, True)]
if os.environ.get("LN_ROWS"):
    CASES = [(int(r), 0.0, True) for r in os.environ["LN_ROWS"].split(",")]
for rows, p, dzin in CASES:
    cols = 768
    g0 = torch.Generator(device="cpu").manual_seed(1)
    mk = lambda *s: torch.randn(*s, generator=g0).bfloat16().to(dev)
    x, res, dy = mk(rows, cols), mk(rows, cols), mk(rows, cols)
    dz = mk(rows, cols) if dzin else None
    bias, g, be = mk(cols), mk(cols), mk(cols)
    so.valor_ln_set_variant(1)
    z, y, mean, rstd = K.bdrln_fwd(x, bias, res, g, be, 1e-5, p_drop=p, seed=1, offset=7)
    outs = {}
    for v in (1, 2, 3):
        so.valor_ln_set_variant(v)
        dx, dres, dg, dbeta, dbias = K.bdrln_bwd(dy, dz, z, mean, rstd, g, p_drop=p, seed=1, offset=7, want_dbias=True)
        outs[v] = [t.float().clone() if t is not None else None for t in (dx, dres, dg, dbeta, dbias)]
    for a, b in zip(outs[2][:2], outs[3][:2]):
        assert (a is None and b is None) or torch.equal(a, b)
    for v in (2, 3):
        for a, b in zip(outs[v], outs[1]):
            if a is not None:
                assert float((a - b).norm() / b.norm().clamp_min(1e-20)) < 4e-3

    def run(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            K.bdrln_bwd(dy, dz, z, mean, rstd, g, p_drop=p, seed=1, offset=7, want_dbias=True)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t = {1: [], 2: [], 3: []}
    for r in range(5):
        for v in (1, 2, 3):
            so.valor_ln_set_variant(v)
            run(2)
            t[v].append(run(10))
    nb = rows * cols * 2 * ((3 if dzin else 2) + (2 if p > 0 else 1))
    row = {f"variant{v}": dict(us=round(sorted(t[v])[2], 1), min_us=round(min(t[v]), 1), TBps=round(nb / sorted(t[v])[2] / 1e6, 2)) for v in (1, 2, 3)}
    res_all[f"{rows}x{cols}_p{p}_dzin{int(dzin)}"] = row
    print(rows, p, dzin, row, flush=True)
so.valor_ln_set_variant(1)
if len(sys.argv) > 1:
    json.dump(res_all, open(sys.argv[1], "w"), indent=1)
