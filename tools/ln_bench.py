"""micro-benchmark of the fused bias+dropout+residual+LayerNorm kernels on the ViT activation shape"""
import sys
import torch
sys.path.insert(0, ".")
from valor_amd import kernels as K
dev = torch.device("cuda:0")
for rows, cols, p in [(100864, 768, 0.0), (16512, 768, 0.1), (8832, 768, 0.1)]:
    x = torch.randn((rows, cols), device=dev).bfloat16(); res = torch.randn_like(x); bias = torch.randn(cols, device=dev).bfloat16()
    g = torch.randn(cols, device=dev).bfloat16(); be = torch.randn(cols, device=dev).bfloat16()
    dy = torch.randn_like(x); dz = torch.randn_like(x)
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
    run(3)
    f, b = min(run(10) for _ in range(3))
    nb = rows * cols * 2
    print(f"rows={rows} p={p}: fwd {f:.1f} us ({4*nb/f/1e6:.2f} TB/s of 4 arrays)  bwd(+3 finalize) {b:.1f} us ({(4 if p == 0 else 5)*nb/b/1e6:.2f} TB/s)", flush=True)
