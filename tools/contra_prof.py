"""The contrastive block (tva / tv / ta fine-grained groups, forward + backward) on synthetic gathered features at a global batch, for
a rocprofv3 --kernel-trace pass. usage: python tools/contra_prof.py [B=512] [reps=3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from valor_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
D, T, F, A = 512, 32, 8, 2
mk = lambda n: torch.nn.functional.normalize(torch.randn(B, n, D, generator=g), dim=-1).to(dev, torch.bfloat16).requires_grad_(True)
ft, fv, fa = mk(T), mk(F), mk(A)
wt, wv, wa = (torch.randn(B, n, generator=g).to(dev).requires_grad_(True) for n in (T, F, A))
maskA = (torch.rand(B, T, generator=g) < 0.7).float().to(dev); maskA[:, 0] = 1
ones = lambda f: torch.ones(f.shape[:2], dtype=torch.float32, device=dev)
k = torch.tensor(14.3, device=dev, requires_grad=True)
for _ in range(reps):
    fB, wB = torch.cat((fv, fa), dim=1), torch.cat((wv, wa), dim=1)
    ls = [ops.fine_contrastive(ft, fB, wt, wB, maskA, ones(fB), k), ops.fine_contrastive(ft, fv, wt, wv, maskA, ones(fv), k),
          ops.fine_contrastive(ft, fa, wt, wa, maskA, ones(fa), k)]
    (sum(ls) / 3).backward()
torch.cuda.synchronize()
