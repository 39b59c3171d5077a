"""micro-benchmark + A/B of the fused bias+dropout+residual+LayerNorm kernel families (valor_ln_set_variant: 0 = one wave per row,
1 = half a wave per row with 16-byte accesses) on the VALOR-base activation shapes; both families are checked against each other
(forward outputs, statistics, backward outputs, column partial sums) before they are timed.
usage: python tools/ln_bench.py [out.json]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from valor_amd import kernels as K, lib  # noqa: E402

dev = torch.device("cuda:0")
so = lib.load()
res_all = {}
for rows, cols, p in [(100864, 768, 0.0), (16512, 768, 0.1), (8832, 768, 0.1), (2049, 512, 0.0), (100353, 1024, 0.1)]:
    x = torch.randn((rows, cols), device=dev).bfloat16(); res = torch.randn_like(x); bias = torch.randn(cols, device=dev).bfloat16()
    g = torch.randn(cols, device=dev).bfloat16(); be = torch.randn(cols, device=dev).bfloat16()
    dy = torch.randn_like(x); dz = torch.randn_like(x)
    outs = {}
    for v in (0, 1):
        so.valor_ln_set_variant(2 * v)
        z, y, mean, rstd = K.bdrln_fwd(x, bias, res, g, be, 1e-5, p_drop=p, seed=1, offset=7)
        dx, dres, dg, dbeta, dbias = K.bdrln_bwd(dy, dz, z, mean, rstd, g, p_drop=p, seed=1, offset=7, want_dbias=True)
        outs[v] = [t.float().clone() for t in (z, y, mean, rstd, dx, dres, dg, dbeta, dbias)]
    names = ("z", "y", "mean", "rstd", "dx", "dres", "dgamma", "dbeta", "dbias")
    errs = {n: float((a - b).norm() / b.norm().clamp_min(1e-20)) for n, a, b in zip(names, outs[1], outs[0])}
    assert torch.equal(outs[0][0], outs[1][0]), "z must be bit-identical (same dropout windows, same adds)"
    assert max(errs.values()) < 4e-3, errs

    def run(n):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        for _ in range(n):
            z, y, mean, rstd = K.bdrln_fwd(x, bias, res, g, be, 1e-5, p_drop=p, seed=1, offset=0)
        e[1].record()
        for _ in range(n):
            K.bdrln_bwd(dy, dz, z, mean, rstd, g, p_drop=p, seed=1, offset=0, want_dbias=True)
        e[2].record(); torch.cuda.synchronize()
        return e[0].elapsed_time(e[1]) / n * 1e3, e[1].elapsed_time(e[2]) / n * 1e3
    nb = rows * cols * 2
    line = {}
    best = {0: [1e9, 1e9], 1: [1e9, 1e9]}
    for r in range(3):
        for v in (0, 1):
            so.valor_ln_set_variant(2 * v)
            run(2)
            f, b = run(10)
            best[v] = [min(best[v][0], f), min(best[v][1], b)]
    for v in (0, 1):
        f, b = best[v]
        line[f"variant{v}"] = dict(fwd_us=round(f, 1), fwd_TBps=round(4 * nb / f / 1e6, 2), bwd_us=round(b, 1), bwd_TBps=round((4 if p == 0 else 5) * nb / b / 1e6, 2))
    res_all[f"{rows}x{cols}_p{p}"] = dict(line, max_rel_diff=max(errs.values()))
    print(f"rows={rows} cols={cols} p={p}: " + "  ".join(f"v{v}: fwd {best[v][0]:.1f} us ({4*nb/best[v][0]/1e6:.2f} TB/s) bwd+finalize {best[v][1]:.1f} us" for v in (0, 1)), flush=True)
so.valor_ln_set_variant(1)
if len(sys.argv) > 1:
    json.dump(res_all, open(sys.argv[1], "w"), indent=1)
