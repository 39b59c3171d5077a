"""(The key-10 half of this tool measured a path that was removed afterwards -- see profiles/LOG.md; the numbers it wrote are in
profiles/r05_gemm_mfma32_ab.json. Policy key 10 is a no-op now.)
A/B of the family-4 GEMM's two round-5 additions at the step's FORWARD (NN) shapes, all forced onto family 4 (policy key 8 = 1):
  key 9:  main loop on v_mfma_f32_32x32x16_bf16 (1) against v_mfma_f32_16x16x32_bf16 (0), both schedules;
  key 10: the forward that saves act'(u) beside act(u) through the bf16 half-tile epilogue (1) against the general fp32 epilogue (0).
Same random bf16 operands, interleaved rounds in one process, HIP events, median of the rounds; TF/s = 2 M N K / time.
usage: python tools/gemm_mfma32_ab.py out.json"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K, lib  # noqa: E402

M, MKV, W, I = 100864, 117376, 768, 3072
SHAPES = [("vit_fc1_fwd", M, I, W), ("vit_qkv_fwd", M, 3 * W, W), ("kv_proj", MKV, 2 * W, W), ("vit_proj_fwd", M, W, W), ("vit_fc2_fwd", M, W, I),
          ("ast_fc1_fwd", 16512, I, W), ("ast_qkv_fwd", 16512, 3 * W, W), ("ast_fc2_fwd", 16512, W, I), ("dec_fc1_fwd", 8832, I, W),
          ("dec_qkv_fwd", 8832, 3 * W, W), ("dec_fc2_fwd", 8832, W, I), ("vocab_logits", 2100, 30522, W)]


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
    so.valor_gemm_set_policy(8, 1)
    res = {"narrow_workgroups_per_cu": so.valor_gemm_narrow_occupancy(), "shapes": {}}
    for name, m, n, k in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(1)
        A = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
        B = (0.05 * torch.randn((n, k), generator=g)).to(torch.bfloat16).to(dev)
        bias = torch.randn(n, generator=g).to(torch.bfloat16).to(dev)
        out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        if n % 8:
            out = torch.empty((m, (n + 7) // 8 * 8), dtype=torch.bfloat16, device=dev)[:, :n]
        assert so.valor_gemm_kernel_for(0, 0, 0, m, n, k, 0) == 4, name
        plain = lambda: K.gemm(A, B, out=out)
        two = lambda: K.gemm(A, B, bias=bias, act=lib.ACT_QUICK_GELU | lib.ACT_DERIV, want_preact=True)
        t = {}
        for rnd in range(3):
            for sched in (1, 0):
                so.valor_gemm_set_narrow_sched(sched)
                for m32 in (0, 1):
                    so.valor_gemm_set_policy(9, m32)
                    t.setdefault(f"plain_s{sched}_m32={m32}", []).append(timeit(plain))
            so.valor_gemm_set_narrow_sched(1)
            if n >= 1536 and m >= 8832:
                for m32 in (0, 1):
                    so.valor_gemm_set_policy(9, m32)
                    for two_out in (0, 1):
                        so.valor_gemm_set_policy(10, two_out)
                        t.setdefault(f"act+deriv_m32={m32}_tile={two_out}", []).append(timeit(two))
                so.valor_gemm_set_policy(10, 1)
            so.valor_gemm_set_policy(9, 0)
        fl = 2.0 * m * n * k
        row = {kk: {"us": round(statistics.median(v), 1), "tflops": round(fl / statistics.median(v) / 1e6, 1)} for kk, v in t.items()}
        res["shapes"][f"{name} {m}x{n}x{k}"] = row
        print(name, {kk: (v["us"], v["tflops"]) for kk, v in row.items()}, flush=True)
        del A, B, out
    with open(sys.argv[1], "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
