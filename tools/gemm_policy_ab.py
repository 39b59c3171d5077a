"""A/B of valor_gemm launch policies on the VALOR-base GEMM shapes, configurations interleaved in one process (HIP events,
min over rounds): kernel family (128x128 LDS-DMA vs 256x256 8-phase) and the 8-phase kernel's epilogue (general fp32 two-pass vs bf16 tile passes).
(Round-2 session A also measured a start skew of the first round's workgroups: slower everywhere, profiles/r02_gemm_policy_ab.json.)
Every configuration is first checked against fp64 torch on one shape (a policy may change speed, never results).
usage: python tools/gemm_policy_ab.py [out.json]"""
import json
import sys

import torch

sys.path.insert(0, ".")
from valor_amd import kernels as K, lib  # noqa: E402

dev = torch.device("cuda:0")
so = lib.load()

# name -> (variant, fast_epilogue, skew units)
CONFIGS = {
    "128sq": (1, 0, 0),
    "8ph_general_epi": (3, 0, 0),
    "8ph_tile_epi": (3, 1, 0),
}

b = 64
T = b * 8 * 197
SHAPES = [  # name, M, N, K, ta, tb, epilogue ("" plain | "bias" | "gelu": bias + QuickGELU + pre-activation copy | "dact": * act'(aux) | "*_deriv": the ACT_DERIV pair)
    ("vit_qkv_fwd", T, 2304, 768, 0, 0, "bias"),
    ("vit_fc1_fwd", T, 3072, 768, 0, 0, "gelu"),
    ("vit_fc1_fwd_plain", T, 3072, 768, 0, 0, ""),
    ("vit_proj_fwd", T, 768, 768, 0, 0, ""),
    ("vit_fc2_fwd", T, 768, 3072, 0, 0, ""),
    ("vit_fc2_dgrad", T, 3072, 768, 0, 1, ""),
    ("vit_fc2_dgrad_dact", T, 3072, 768, 0, 1, "dact"),
    ("vit_fc1_fwd_deriv", T, 3072, 768, 0, 0, "gelu_deriv"),
    ("vit_fc2_dgrad_dderiv", T, 3072, 768, 0, 1, "dact_deriv"),
    ("vit_proj_dgrad", T, 768, 768, 0, 1, ""),
    ("vit_qkv_dgrad", T, 768, 2304, 0, 1, ""),
    ("vit_fc1_dgrad", T, 768, 3072, 0, 1, ""),
    ("ast_fc1_fwd", b * 2 * 129, 3072, 768, 0, 0, "gelu"),
    ("ast_fc2_dgrad", b * 2 * 129, 3072, 768, 0, 1, "dact"),
    ("ast_fc1_fwd_deriv", b * 2 * 129, 3072, 768, 0, 0, "gelu_deriv"),
    ("ast_fc2_dgrad_dderiv", b * 2 * 129, 3072, 768, 0, 1, "dact_deriv"),
    ("ast_proj_fwd", b * 2 * 129, 768, 768, 0, 0, ""),
    ("xkv_fwd", b * 1834, 1536, 768, 0, 0, "bias"),
    ("xkv_dgrad", b * 1834, 768, 1536, 0, 1, ""),
    ("dec_fc1_fwd", 8832, 3072, 768, 0, 0, "gelu"),
    ("dec_fc2_dgrad", 8832, 3072, 768, 0, 1, "dact"),
]


def mk(shape, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(shape, generator=g).to(torch.bfloat16).to(dev)


def apply(cfg):
    v, fe, sk = CONFIGS[cfg]
    so.valor_gemm_set_variant(v)
    so.valor_gemm_set_fast_epilogue(fe)
    so.valor_gemm_set_policy(1, sk)


def run(A, B, ta, tb, epi, bias, out, pre):
    if epi in ("gelu", "gelu_deriv"):
        act = lib.ACT_QUICK_GELU | (lib.ACT_DERIV if epi == "gelu_deriv" else 0)
        lib.call("valor_gemm", K._stream(), 0, ta, tb, out.shape[0], out.shape[1], A.shape[0] if ta else A.shape[1], A.data_ptr(), A.stride(0),
                 B.data_ptr(), B.stride(0), out.data_ptr(), out.stride(0), bias.data_ptr(), act, pre.data_ptr(), 0, 0, 1.0, 0, 0, 0, 0, 0, 0)
    elif epi in ("dact", "dact_deriv"):
        K.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), act=lib.ACT_QUICK_GELU | (lib.ACT_DERIV if epi == "dact_deriv" else 0), dact_aux=pre, out=out, splitk=False)
    else:
        K.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), bias=bias if epi == "bias" else None, out=out, splitk=False)


def check():
    M, N, Kd = 1000, 776, 768
    A, Bn, Bt, bias = mk((M, Kd), 1), mk((N, Kd), 2), mk((Kd, N), 3), mk((N,), 4)
    refn = A.double() @ Bn.double().t() + bias.double()
    reft = A.double() @ Bt.double()
    worst = {}
    for cfg in CONFIGS:
        apply(cfg)
        c1 = K.gemm(A, Bn, bias=bias, splitk=False)
        c2 = K.gemm(A, Bt, trans_b=True, splitk=False)
        e = max(float((c1.double() - refn).norm() / refn.norm()), float((c2.double() - reft).norm() / reft.norm()))
        worst[cfg] = e
        assert e < 6e-3, (cfg, e)
    print("check:", {k: f"{v:.2e}" for k, v in worst.items()}, flush=True)


def bench(rounds=4, n=6):
    res = {}
    for name, M, N, Kd, ta, tb, epi in SHAPES:
        A = mk((Kd, M) if ta else (M, Kd), 11)
        B = mk((Kd, N) if tb else (N, Kd), 12)
        bias = mk((N,), 13)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        pre = mk((M, N), 14) if epi in ("gelu", "dact", "gelu_deriv", "dact_deriv") else None
        best = {c: 1e9 for c in CONFIGS}
        for r in range(rounds):
            for cfg in CONFIGS:
                apply(cfg)
                run(A, B, ta, tb, epi, bias, out, pre)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    run(A, B, ta, tb, epi, bias, out, pre)
                e1.record()
                torch.cuda.synchronize()
                best[cfg] = min(best[cfg], e0.elapsed_time(e1) / n)
        fl = 2.0 * M * N * Kd
        res[name] = {c: {"us": round(best[c] * 1e3, 1), "TF": round(fl / best[c] / 1e9, 1)} for c in CONFIGS}
        print(f"{name:18s} M={M:7d} N={N:5d} K={Kd:5d} " + "  ".join(f"{c}: {best[c]*1e3:6.1f}us {fl / best[c] / 1e9:6.0f}TF" for c in CONFIGS), flush=True)
        del A, B, out, pre
    return res


if __name__ == "__main__":
    check()
    r = bench()
    apply("128sq"); so.valor_gemm_set_variant(4)
    if len(sys.argv) > 1:
        json.dump({"configs": {k: dict(variant=v[0], fast_epilogue=v[1], skew=v[2]) for k, v in CONFIGS.items()}, "results": r},
                  open(sys.argv[1], "w"), indent=1)
