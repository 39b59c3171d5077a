"""debug: native fp32 gradients against the CPU oracle's at the bench batch (B = 64 golden recipe) -- which tensors / rows differ"""
import os, random, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T
g = torch.load(os.path.join(ROOT, "tests/golden/ref_base_b64f8a2_q.pt"), weights_only=False)
rc = g["recipe"]
B = int(os.environ.get("DBG_B", rc["batch"]))
rc = dict(rc, batch=B)
task = os.environ.get("DBG_TASK", rc["task"])
spec, sd, batch = T._recipe_tensors(rc)
dev = torch.device("cuda:0")
model = T._native(spec, sd, torch.float32, dev)
random.seed(rc["masker_seed"])
out = model(batch, task=task, compute_loss=True)
sum(out.values()).backward()
torch.cuda.synchronize()
ng = {k: v.detach().float().cpu() for k, v in T._native_grads(model).items()}
print("native losses", {k: float(v) for k, v in out.items()}, flush=True)
print(open("/proc/meminfo").read().split("\n")[0], flush=True)
t0 = time.time()
orc, sd_o = T._oracle(spec, sd)
random.seed(rc["masker_seed"])
oo = orc.forward_pt(batch, task, compute_loss=True)
sum(oo.values()).backward()
print("oracle losses", {k: float(v) for k, v in oo.items()}, "in", round(time.time() - t0, 1), "s", flush=True)
import valor_oracle as VO
rows = []
for k, p in sd_o.items():
    if VO.is_alias_key(k) or not p.is_floating_point() or p.grad is None or k not in ng:
        continue
    go, gn = p.grad, ng[k].reshape(p.grad.shape)
    rows.append((float((gn - go).norm() / go.norm().clamp_min(1e-12)), k, float(go.norm()), float(gn.norm())))
rows.sort(reverse=True)
for r in rows[:15]:
    print("%.3e  %s  oracle %.6g native %.6g" % r)
k = "multimodal_encoder.embeddings.word_embeddings.weight"
go, gn = sd_o[k].grad, ng[k]
d = (gn - go).norm(dim=1)
top = d.topk(10)
tok = batch["txt_tokens"]["bert_tokens"]
print("rows with the largest difference:", [(int(i), float(v), float(go[i].norm()), float(gn[i].norm()), int((tok == i).sum())) for v, i in zip(top.values, top.indices)])
if k in g["steps"][0]["grad_norm"] and B == g["recipe"]["batch"] and task == g["recipe"]["task"]:
    print("reference norm", g["steps"][0]["grad_norm"][k], "oracle", float(go.norm()), "native", float(gn.norm()))
