"""A/B of the valor_gemm bf16 kernel variants: correctness of every layout (incl. tails) against fp64 torch and
HIP-event timings on the VALOR-base GEMM shapes, variants interleaved in one process.
usage: python tools/gemm_ab.py [out.json] [variants e.g. 0,1,2]"""
import itertools
import json
import sys

import torch

sys.path.insert(0, ".")
from valor_amd import kernels as K, lib  # noqa: E402

dev = torch.device("cuda:0")
so = lib.load()
VARS = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "1,3".split(","))]


def mk(shape, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(shape, generator=g).to(torch.bfloat16).to(dev)


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / max(b.norm().item(), 1e-6)).item()


def check():
    shapes = [(128, 128, 64), (256, 384, 128), (200, 136, 72), (8, 8, 8), (304, 1912, 768), (4104, 96, 40), (136, 257 * 8, 1000), (1000, 264, 4160)]
    worst = {}
    for v in VARS:
        so.valor_gemm_set_variant(v)
        w = 0.0
        for (M, N, Kd), ta, tb in itertools.product(shapes, [False, True], [False, True]):
            A = mk((Kd, M) if ta else (M, Kd), 1)
            B = mk((Kd, N) if tb else (N, Kd), 2)
            ref = (A.t() if ta else A).double() @ (B if tb else B.t()).double()
            for sk in (False, True):
                C = K.gemm(A, B, trans_a=ta, trans_b=tb, splitk=sk)
                e = rel(C, ref)
                w = max(w, e)
                if e > 6e-3:
                    print(f"FAIL variant {v} M={M} N={N} K={Kd} ta={ta} tb={tb} splitk={sk} rel={e:.3e}", flush=True)
        # k tail with NaN pads + padded ld (direct operands), and a sub-view A operand
        M, N, Kd, ldk = 70, 136, 1001, 1008
        Ab, Bb = mk((M, ldk), 3), mk((N, ldk), 4)
        Ab[:, Kd:] = float("nan"); Bb[:, Kd:] = float("nan")
        C = K.gemm(Ab[:, :Kd], Bb[:, :Kd], splitk=False)
        e = rel(C, Ab[:, :Kd].double() @ Bb[:, :Kd].double().t()); w = max(w, e)
        Bt = mk((Kd, 136), 5)
        C2 = K.gemm(Ab[:, :Kd], Bt, trans_b=True, splitk=False)
        e2 = rel(C2, Ab[:, :Kd].double() @ Bt.double()); w = max(w, e2)
        # epilogue: bias + gelu + preact, dact, fp32 accumulate
        X, W, b = mk((300, 256), 6), mk((520, 256), 7), mk((520,), 8)
        out, pre = K.gemm(X, W, bias=b, act=lib.ACT_GELU_ERF, want_preact=True, splitk=False)
        pr = X.double() @ W.double().t() + b.double()
        e3 = max(rel(pre, pr), rel(out, torch.nn.functional.gelu(pr))); w = max(w, e3)
        print(f"variant {v}: worst rel err {w:.3e}  (ktail {e:.2e} {e2:.2e}, epilogue {e3:.2e})", flush=True)
        worst[v] = w
    return worst


b = 64
SHAPES = [
    ("vit_qkv_fwd", b * 8 * 197, 2304, 768, 0, 0),
    ("vit_fc1_fwd", b * 8 * 197, 3072, 768, 0, 0),
    ("vit_fc2_fwd", b * 8 * 197, 768, 3072, 0, 0),
    ("vit_proj_fwd", b * 8 * 197, 768, 768, 0, 0),
    ("vit_fc1_dgrad", b * 8 * 197, 768, 3072, 0, 1),
    ("vit_fc2_dgrad", b * 8 * 197, 3072, 768, 0, 1),
    ("vit_qkv_dgrad", b * 8 * 197, 768, 2304, 0, 1),
    ("vit_fc1_wgrad", 3072, 768, b * 8 * 197, 1, 1),
    ("vit_fc2_wgrad", 768, 3072, b * 8 * 197, 1, 1),
    ("vit_qkv_wgrad", 2304, 768, b * 8 * 197, 1, 1),
    ("vit_proj_wgrad", 768, 768, b * 8 * 197, 1, 1),
    ("xkv_fwd", b * 1834, 1536, 768, 0, 0),
    ("xkv_wgrad", 1536, 768, b * 1834, 1, 1),
    ("ast_fc1_fwd", b * 2 * 129, 3072, 768, 0, 0),
    ("dec_fc1_fwd", 3 * b * 32, 3072, 768, 0, 0),
    ("dec_fc1_wgrad", 3072, 768, 3 * b * 32, 1, 1),
    ("head_decoder", 2400, 30522, 768, 0, 0),
]
SMALL = [
    ("dec_qkv_fwd", 6144, 2304, 768, 0, 0),
    ("dec_proj_fwd", 6144, 768, 768, 0, 0),
    ("dec_fc2_fwd", 6144, 768, 3072, 0, 0),
    ("mlm_proj_fwd", 2688, 768, 768, 0, 0),
    ("mlm_fc1_fwd", 2688, 3072, 768, 0, 0),
    ("mlm_fc2_fwd", 2688, 768, 3072, 0, 0),
    ("dec_proj_dgrad", 6144, 768, 768, 0, 1),
    ("dec_fc1_dgrad", 6144, 768, 3072, 0, 1),
    ("dec_fc2_dgrad", 6144, 3072, 768, 0, 1),
    ("mlm_fc2_dgrad", 2688, 3072, 768, 0, 1),
    ("dec_proj_wgrad", 768, 768, 6144, 1, 1),
    ("dec_fc1_wgrad", 3072, 768, 6144, 1, 1),
    ("mlm_proj_wgrad", 768, 768, 2688, 1, 1),
    ("mlm_fc1_wgrad", 3072, 768, 2688, 1, 1),
    ("txt_qkv_fwd", 2048, 1536, 512, 0, 0),
    ("ast_proj_fwd", 16512, 768, 768, 0, 0),
    ("ast_fc2_dgrad", 16512, 3072, 768, 0, 1),
]
if len(sys.argv) > 3 and sys.argv[3] == "small":
    SHAPES = SMALL


def bench(rounds=3, n=8):
    res = {}
    ops = []
    for name, M, N, Kd, ta, tb in SHAPES:
        A = mk((Kd, M) if ta else (M, Kd), 11)
        B = mk((Kd, N) if tb else (N, Kd), 12)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ops.append((name, M, N, Kd, ta, tb, A, B, out))
    for name, M, N, Kd, ta, tb, A, B, out in ops:
        best = {v: 1e9 for v in VARS}
        for r in range(rounds):
            for v in VARS:
                so.valor_gemm_set_variant(v)
                K.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    K.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=out)
                e1.record()
                torch.cuda.synchronize()
                best[v] = min(best[v], e0.elapsed_time(e1) / n)
        fl = 2.0 * M * N * Kd
        res[name] = {str(v): round(fl / best[v] / 1e9, 1) for v in VARS}
        print(f"{name:16s} M={M:7d} N={N:6d} K={Kd:7d} " + "  ".join(f"v{v}: {best[v]*1e3:7.1f} us {fl / best[v] / 1e9:7.1f} TF" for v in VARS), flush=True)
    return res


if __name__ == "__main__":
    w = check()
    r = bench()
    if len(sys.argv) > 1:
        json.dump({"worst_rel_err": w, "TF": r}, open(sys.argv[1], "w"), indent=1)
