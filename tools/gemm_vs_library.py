"""Ceiling reference (cdna_hip_programming.md rule 10: a ceiling claim needs a known-good reference measured on the same hardware):
the step's big GEMM shapes on valor_gemm (default policy) and on torch.matmul (rocBLAS / hipBLASLt through PyTorch-ROCm), same random
bf16 operands, interleaved rounds, HIP events. The library call is plain (no bias / activation / second output) -- it is only timed
here, nothing under valor_amd/ links it. usage: python tools/gemm_vs_library.py out.json [small]
"small": the decoder (8832 token rows) and AST (16512) shapes in all three orientations -- the launches that fill less than a round or two of
tiles and run on the 128 x 128 kernels or on many K-slices."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K, lib  # noqa: E402

M, W, I = 100864, 768, 3072
SHAPES = [("fc1_fwd NN", M, I, W, 0, 0), ("qkv_fwd NN", M, 3 * W, W, 0, 0), ("fc2_fwd NN", M, W, I, 0, 0), ("proj_fwd NN", M, W, W, 0, 0),
          ("kv_proj NN", 117376, 2 * W, W, 0, 0), ("fc2_dgrad NT", M, I, W, 0, 1), ("fc1_dgrad NT", M, W, I, 0, 1),
          ("fc1_wgrad TT", I, W, M, 1, 1), ("fc2_wgrad TT", W, I, M, 1, 1), ("qkv_wgrad TT", 3 * W, W, M, 1, 1),
          ("ast_fc1 NN", 16512, I, W, 0, 0), ("dec_fc1 NN", 8832, I, W, 0, 0), ("dec_qkv NN", 8832, 3 * W, W, 0, 0)]




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
    res = {}
    for name, m, n, k, ta, tb in (small_shapes() if len(sys.argv) > 2 and sys.argv[2] == "small" else SHAPES):
        g = torch.Generator(device="cpu").manual_seed(1)
        A = torch.randn((k, m) if ta else (m, k), generator=g).to(torch.bfloat16).to(dev)
        B = (0.05 * torch.randn((k, n) if tb else (n, k), generator=g)).to(torch.bfloat16).to(dev)
        out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        ours = lambda: K.gemm(A, B, trans_a=bool(ta), trans_b=bool(tb), out=out)
        opA, opB = (A.t() if ta else A), (B if tb else B.t())
        lib_ = lambda: torch.matmul(opA, opB, out=out)
        t_o, t_l = [], []
        for _ in range(3):
            t_o.append(timeit(ours)); t_l.append(timeit(lib_))
        fl = 2.0 * m * n * k
        o, l = sorted(t_o)[1], sorted(t_l)[1]
        fam = lib.load().valor_gemm_kernel_for(0, ta, tb, m, n, k, 0)      # 3 = 256 x 256 8-phase, else the 128 x 128 LDS-DMA kernels
        res[name] = {"MNK": [m, n, k], "family": fam, "valor_us": round(o, 1), "valor_TF": round(fl / o / 1e6, 1), "library_us": round(l, 1), "library_TF": round(fl / l / 1e6, 1),
                     "valor_over_library": round(l / o, 3)}
        print(name, res[name], flush=True)
    json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
