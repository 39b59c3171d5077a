"""forward of the fused bias + dropout + residual + LayerNorm kernel at the ViT shape under the env presets of csrc/layernorm.hip
(VALOR_LN_FWD_MODE, VALOR_LN_FWD_BLOCKS: read once per process, so one process per arm -- tools/step_ab.sh style). Prints the time, the
algorithmic TB/s (x + residual read, z + y written) and checksums of z / y / mean / rstd (the arms must agree bit for bit)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
out = []
for rows, cols, p in [(100864, 768, 0.1), (100864, 768, 0.0), (16512, 768, 0.1)]:
    g0 = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((rows, cols), generator=g0).bfloat16().to(dev); res = torch.randn((rows, cols), generator=g0).bfloat16().to(dev)
    bias = torch.randn(cols, generator=g0).bfloat16().to(dev); g = torch.randn(cols, generator=g0).bfloat16().to(dev); be = torch.randn(cols, generator=g0).bfloat16().to(dev)
    z, y, mean, rstd = K.bdrln_fwd(x, bias, res, g, be, 1e-5, p_drop=p, seed=1, offset=7)
    cs = [float(t.double().sum()) for t in (z, y, mean, rstd)]
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.bdrln_fwd(x, bias, res, g, be, 1e-5, p_drop=p, seed=1, offset=0)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    out.append(f"{rows}x{cols} p={p}: {best:.1f} us {4 * rows * cols * 2 / best / 1e6:.2f} TB/s cs={cs[0]:.6e},{cs[1]:.6e},{cs[2]:.6e},{cs[3]:.6e}")
# backward at the ViT shape under valor_ln_set_nt bit 5 (dz_in requested with z / dy)
from valor_amd import lib  # noqa: E402
so = lib.load()
rows, cols, p = 100864, 768, 0.1
g0 = torch.Generator(device="cpu").manual_seed(2)
z = torch.randn((rows, cols), generator=g0).bfloat16().to(dev); dy = torch.randn((rows, cols), generator=g0).bfloat16().to(dev); dz = torch.randn((rows, cols), generator=g0).bfloat16().to(dev)
mean = torch.randn(rows, generator=g0).to(dev); rstd = torch.rand(rows, generator=g0).to(dev) + 0.5; g = torch.randn(cols, generator=g0).bfloat16().to(dev)
ref = None
for bit in (0, 32, 0, 32):
    so.valor_ln_set_nt(bit)
    r = K.bdrln_bwd(dy, dz, z, mean, rstd, g, p_drop=p, seed=1, offset=3, want_dbias=True)
    cs = [float(t.double().sum()) for t in r if t is not None]
    ref = ref or cs
    assert cs == ref, (cs, ref)
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.bdrln_bwd(dy, dz, z, mean, rstd, g, p_drop=p, seed=1, offset=3, want_dbias=True)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    out.append(f"bwd nt={bit}: {best:.1f} us")
so.valor_ln_set_nt(0)
print(" | ".join(out))
