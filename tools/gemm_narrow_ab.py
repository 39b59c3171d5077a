"""A/B of the 256 x 128 two-workgroups-per-CU 8-phase GEMM (kernel family 4, csrc/gemm8n.hip) against what the policy runs without it
(family 3 = 256 x 256 8-phase, family 1 = 128 x 128 LDS-DMA) and against torch.matmul (rocBLAS / hipBLASLt, timed only), at the step's
GEMM shapes: the ViT-sized problems (100 864 / 117 376 rows) and the decoder (8832 rows) / AST (16 512) ones in all three orientations.
Same random bf16 operands, interleaved rounds in one process, HIP events, median of the rounds.
usage: python tools/gemm_narrow_ab.py out.json [big|small|all] [epi]
"epi": the fused-epilogue launches of the step instead (bias + QuickGELU with the derivative copy, saved-derivative dgrad, C +=)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K, lib  # noqa: E402

M, MKV, W, I = 100864, 117376, 768, 3072
BIG = [("vit_fc1_fwd NN", M, I, W, 0, 0), ("vit_qkv_fwd NN", M, 3 * W, W, 0, 0), ("vit_fc2_fwd NN", M, W, I, 0, 0), ("vit_proj_fwd NN", M, W, W, 0, 0),
       ("kv_proj NN", MKV, 2 * W, W, 0, 0), ("vit_fc2_dgrad NT", M, I, W, 0, 1), ("vit_fc1_dgrad NT", M, W, I, 0, 1), ("vit_qkv_dgrad NT", M, W, 3 * W, 0, 1),
       ("vit_proj_dgrad NT", M, W, W, 0, 1), ("kv_dgrad NT", MKV, W, 2 * W, 0, 1),
       ("vit_fc1_wgrad TT", I, W, M, 1, 1), ("vit_fc2_wgrad TT", W, I, M, 1, 1), ("vit_qkv_wgrad TT", 3 * W, W, M, 1, 1), ("vit_proj_wgrad TT", W, W, M, 1, 1)]


def small_shapes():
    out = []
    for tag, m in (("dec", 8832), ("ast", 16512)):
        for nm, n, k in (("qkv", 3 * W, W), ("proj", W, W), ("fc1", I, W), ("fc2", W, I)):
            out.append((f"{tag}_{nm}_fwd NN", m, n, k, 0, 0))
            out.append((f"{tag}_{nm}_dgrad NT", m, k, n, 0, 1))
            out.append((f"{tag}_{nm}_wgrad TT", n, k, m, 1, 1))
    return out


def timeit(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda", 0)
    so = lib.load()
    which = sys.argv[2] if len(sys.argv) > 2 else "all"
    epi = len(sys.argv) > 3 and sys.argv[3] == "epi"
    shapes = (BIG if which in ("big", "all") else []) + (small_shapes() if which in ("small", "all") else [])
    res = {"narrow_workgroups_per_cu": so.valor_gemm_narrow_occupancy()}
    print(res, flush=True)
    for name, m, n, k, ta, tb in shapes:
        g = torch.Generator(device="cpu").manual_seed(1)
        A = torch.randn((k, m) if ta else (m, k), generator=g).to(torch.bfloat16).to(dev)
        B = (0.05 * torch.randn((k, n) if tb else (n, k), generator=g)).to(torch.bfloat16).to(dev)
        out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        kw = {}
        if epi:
            if ta:
                continue
            if not tb:      # forward: bias + QuickGELU, derivative copy
                kw = dict(bias=torch.randn(n, generator=g).to(torch.bfloat16).to(dev), act=lib.ACT_QUICK_GELU | lib.ACT_DERIV, want_preact=True)
                ours = lambda: K.gemm(A, B, **kw)
            else:           # dgrad: multiply by the saved derivative, C +=
                D = torch.rand((m, n), generator=g).to(torch.bfloat16).to(dev)
                kw = dict(act=lib.ACT_QUICK_GELU | lib.ACT_DERIV, dact_aux=D)
                ours = lambda: K.gemm(A, B, trans_b=True, out=out, accumulate=True, **kw)
        else:
            ours = lambda: K.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=out)
        opA, opB = (A.t() if ta else A), (B if tb else B.t())
        lib_ = lambda: torch.matmul(opA, opB, out=out)
        fam0 = so.valor_gemm_kernel_for(0, ta, tb, m, n, k, 0)
        so.valor_gemm_set_policy(8, 1)
        fam4 = so.valor_gemm_kernel_for(0, ta, tb, m, n, k, 0)
        so.valor_gemm_set_policy(8, 0)
        t = {"base": [], "base_p": [], "n0": [], "n1": [], "lib": []}
        for _ in range(3):
            so.valor_gemm_set_policy(8, 0)
            so.valor_gemm_set_8ph_sched(0)
            t["base"].append(timeit(ours))
            if fam0 == 3:                       # the 256 x 256 kernel with its pipelined K loop
                so.valor_gemm_set_8ph_sched(1)
                t["base_p"].append(timeit(ours))
                so.valor_gemm_set_8ph_sched(0)
            if fam4 == 4:
                so.valor_gemm_set_policy(8, 1)
                if os.environ.get("AB_PLAIN"):
                    so.valor_gemm_set_narrow_sched(0)
                    t["n0"].append(timeit(ours))
                so.valor_gemm_set_narrow_sched(1)
                t["n1"].append(timeit(ours))
                so.valor_gemm_set_narrow_sched(0)
                so.valor_gemm_set_policy(8, 0)
            if not epi:
                t["lib"].append(timeit(lib_))
        fl = 2.0 * m * n * k
        med = {kk: (sorted(v)[len(v) // 2] if v else None) for kk, v in t.items()}
        row = {"MNK": [m, n, k], "base_family": fam0}
        for kk, v in med.items():
            if v:
                row[kk + "_us"] = round(v, 1)
                row[kk + "_TF"] = round(fl / v / 1e6, 1)
        for kk in ("base_p", "n0", "n1"):
            if med[kk]:
                row[kk + "_over_base"] = round(med["base"] / med[kk], 3)
        res[name] = row
        print(name, row, flush=True)
    json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
