import os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T
g = torch.load(os.path.join(ROOT, "tests/golden/ref_base_b64f8a2_q.pt"), weights_only=False)
rc = g["recipe"]
spec, sd, batch = T._recipe_tensors(rc)
dev = torch.device("cuda:0")
K = "multimodal_encoder.embeddings.word_embeddings.weight"
model = T._native(spec, sd, torch.float32, dev)
random.seed(rc["masker_seed"])
out = model(batch, task=rc["task"], compute_loss=True)
sum(out.values()).backward()
torch.cuda.synchronize()
x = T._native_grads(model)[K]
print(x.shape, x.dtype, x.stride(), x.is_contiguous(), x.data_ptr() % 16)
print("gpu norm", float(x.norm()), "gpu sqrt(sum sq)", float((x * x).sum().sqrt()), "gpu double", float(x.double().norm()), "vector_norm", float(torch.linalg.vector_norm(x)))
c = x.cpu()
print("cpu norm", float(c.norm()), "cpu double", float(c.double().norm()))
xc = x.clone()
print("gpu clone norm", float(xc.norm()), "rows norm max", float(x.norm(dim=1).max()), "argmax", int(x.norm(dim=1).argmax()))
print("per-row via gpu:", float((x.norm(dim=1) ** 2).sum().sqrt()))
