"""The static cross-attention K|V buffers of VALOR.project_cross_kv (side-stream projections, outside the caching allocator): one training
forward per backward is the contract -- a second forward rewrites them with raw kernels autograd cannot see, so the nodes that saved them
carry the generation they saw (ops.StaticGen) and must refuse to differentiate through a newer one; eval() releases the pools."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

TASK = "pt_contra%tva%tv%ta_caption%tva%tv%ta_mlm%tva"


def _model(dev):
    from valor_amd import synth
    from valor_amd.model.valor import VALOR
    spec = synth.tiny_spec()
    m = VALOR({"dropout": 0.0, "drop_path_rate": 0.0}, spec=spec, dtype=torch.float32, device=dev)
    m.load_state_dict(synth.make_state_dict(spec, seed=3, w_std=0.05), strict=True)
    m.train()
    return m, synth.make_batch(spec, batch=2, frames=2, audio_slices=1, txt_len=32, seed=4)


def test_second_forward_before_backward_is_refused_and_eval_releases(dev, monkeypatch):
    monkeypatch.setenv("VALOR_ENCODER_STREAMS", "1")
    monkeypatch.setenv("VALOR_KV_STREAM", "1")
    m, batch = _model(dev)
    random.seed(1)
    first = m(batch, task=TASK, compute_loss=True)
    assert m._kv_static is not None and m._kv_gen.gen == 1
    random.seed(2)
    second = m(batch, task=TASK, compute_loss=True)
    assert m._kv_gen.gen == 2
    with pytest.raises(RuntimeError, match="static cross-attention"):
        sum(first.values()).backward()
    m.zero_grad()
    sum(second.values()).backward()                 # the newest graph is the one the buffers belong to
    torch.cuda.synchronize()
    g = m.P["multimodal_encoder.encoder.layer.0.cross_attn.cross.kv.weight"].grad
    assert torch.isfinite(g).all() and float(g.abs().sum()) > 0
    from valor_amd import ops
    mine = list(m._kv_gen.ptrs)
    assert mine and all(q in ops.StaticGen.REGISTRY for q in mine)
    m.eval()
    assert m._kv_static is None and m._dkv_static_pool is None
    assert not any(q in ops.StaticGen.REGISTRY for q in mine)      # (other models of the test session may still hold theirs)
    with torch.no_grad():
        m(batch, task=TASK, compute_loss=True)       # the eval path allocates through the caching allocator
    assert m._kv_static is None


def test_registry_forgets_a_model_that_is_dropped_without_release(dev, monkeypatch):
    import gc
    from valor_amd import ops
    monkeypatch.setenv("VALOR_ENCODER_STREAMS", "1")
    monkeypatch.setenv("VALOR_KV_STREAM", "1")
    m, batch = _model(dev)
    random.seed(1)
    out = m(batch, task=TASK, compute_loss=True)
    mine = list(m._kv_gen.ptrs)
    assert mine and all(q in ops.StaticGen.REGISTRY for q in mine)
    del out, m
    gc.collect()
    assert not any(q in ops.StaticGen.REGISTRY for q in mine)
