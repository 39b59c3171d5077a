"""SURVEY 8 row f1 -- graph capture of the step's launch-bound parts: (a) every dropout entry point reads a device-resident counter
(`rng_base`, include/valor_hip.h) beside its by-value (seed, offset); (b) the CLIP ViT / VideoSwin, AST and CLIP text encoders AND the
decoder's layer stack with its cross-stream K|V projections (a parallel branch inside the graph; token rows and [video | audio] rows
are differentiable graph inputs) replay hipGraphs (valor_amd/graphs.py) with losses and parameters BIT-IDENTICAL to the eager run of
the same seed, fresh dropout masks in every replay, and the gradient writes reported to the data-parallel reducer after every replay. The reference is eager PyTorch (model/pretrain.py:246-263,
train_utils.py:302-364); torch's global generator advancing once per dropout call (train_utils.py:309 loop) is what the per-step
counter replaces."""
import random
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu
TASK = "pt_contra%tva%tv%ta_caption%tva%tv%ta_mlm%tva"


@pytest.fixture
def device_rng(dev):
    from valor_amd import ops
    ops.DropoutState.disable_device_base()
    yield ops.DropoutState
    ops.DropoutState.disable_device_base()
    ops.DropoutState.reset(1234)


def test_device_resident_offset_equals_the_by_value_offset(dev, device_rng):
    """offset (by value) + *rng_base (read by the kernel) is ONE number: a launch with (o, base = B) draws the masks of the launch
    with (o + B, no base) -- in the LayerNorm-side Philox kernels (both row widths' families), the resident / streaming / key-stationary
    attention kernels, forward and backward; a different counter value draws other masks."""
    from valor_amd import kernels as K
    B = 3 * (1 << 40) + 12345

    def ln(rows, cols, off):
        x = torch.ones((rows, cols), dtype=torch.bfloat16, device=dev)
        gam = torch.ones(cols, dtype=torch.bfloat16, device=dev)
        z, y, mean, rstd = K.bdrln_fwd(x, None, torch.zeros_like(x), gam, torch.zeros_like(gam), 1e-5, p_drop=0.3, seed=7, offset=off, write_z=True)
        dy = torch.randn((rows, cols), generator=torch.Generator().manual_seed(2)).to(torch.bfloat16).to(dev)
        dx, dres, *_ = K.bdrln_bwd(dy, None, z, mean, rstd, gam, p_drop=0.3, seed=7, offset=off, want_dgamma=False, want_dbeta=False)
        return z, dx

    def attn(S, Skv, off):
        gq = torch.Generator(device="cpu").manual_seed(9)
        q = torch.randn((4, S, 128), generator=gq).to(torch.bfloat16).to(dev)
        k = torch.randn((4, Skv, 128), generator=gq).to(torch.bfloat16).to(dev)
        v = torch.randn((4, Skv, 128), generator=gq).to(torch.bfloat16).to(dev)
        o, lse = K.attn_fwd(q, k, v, 2, p_drop=0.25, seed=11, offset=off)
        do = torch.ones_like(o)
        dq, dk, dv = K.attn_bwd(q, k, v, o, lse, do, 2, p_drop=0.25, seed=11, offset=off)
        return o, dq, dk, dv

    cases = [lambda off: ln(4096, 768, off), lambda off: ln(1024, 1000, off), lambda off: ln(70000, 768, off),
             lambda off: attn(197, 197, off), lambda off: attn(48, 1834, off), lambda off: attn(300, 300, off)]
    for case in cases:
        want = case(1000 + B)                               # host mode: everything by value
        base = device_rng.enable_device_base(dev)
        base.fill_(B)
        got = case(1000)
        base.fill_(B + (1 << 40))
        other = case(1000)
        device_rng.disable_device_base()
        for a, b in zip(want, got):
            assert torch.equal(a, b)
        assert not torch.equal(other[0], got[0])
        if len(want) == 2:       # LayerNorm side: x = 1, residual = 0 -> z is 0 where dropped, 1 / 0.7 where kept; dx vanishes exactly there
            z, dx = got
            kept = z != 0
            assert 0.68 < float(kept.float().mean()) < 0.72
            assert bool(((dx != 0) <= kept).all()) and float((dx != 0).float().mean()) > 0.6


def _engine(dev, graphs, seed=3, swin=False):
    from valor_amd import ops, synth
    from valor_amd.engine import TrainEngine
    from valor_amd.model.valor import VALOR
    spec = synth.tiny_swin_spec() if swin else synth.tiny_spec()
    sd = synth.make_state_dict(spec, seed=seed, w_std=0.05)
    model = VALOR({"dropout": 0.1, "drop_path_rate": 0.2 if swin else 0.0}, spec=spec, dtype=torch.bfloat16, device=dev)
    model.load_state_dict(sd, strict=True)
    opts = SimpleNamespace(learning_rate=1e-3, weight_decay=0.01, clip_lr=1e-4, clip_lr_text=1e-4, new_lr=0.0, decoder_lr=-1, betas=[0.9, 0.98],
                           warmup_ratio=0.1, num_train_steps=100, scheduler="warmup_linear", grad_norm=5.0, alloc_headroom_mb=0)
    eng = TrainEngine(model, opts, manage_gc=False, graphs=graphs)
    eng.optimizer.init_master_from(sd)
    if not graphs:
        ops.DropoutState.enable_device_base(dev)            # the eager twin draws from the same device-mode windows
    batch = synth.make_batch(spec, batch=4, frames=2, audio_slices=1, txt_len=32, seed=4)
    batch["video_pixels"] = batch["video_pixels"].to(dev)
    batch["audio_spectrograms"] = batch["audio_spectrograms"].to(dev)
    return model, eng, batch


def test_graphed_encoders_are_bit_identical_to_the_eager_step(dev, device_rng):
    """six training steps (forward + backward + clip + AdamW) with the encoders captured on their third call and replayed from then on,
    against six eager steps of the same seed: every loss of every step and every parameter after the last one equal to the bit."""
    from valor_amd import ops
    runs = {}
    for graphs in (False, True):
        model, eng, batch = _engine(dev, graphs)
        ops.DropoutState.reset(77)
        random.seed(5)
        losses = []
        for step in range(6):
            out = eng.train_step(batch, TASK)
            losses.append({k: float(v) for k, v in out.items()})
        torch.cuda.synchronize()
        if graphs:
            segs = model._graph_segs
            assert set(segs) == {"vit", "ast", "clip_text", "decoder"} and all(len(s.captured) == 1 for s in segs.values())
            dec = next(iter(segs["decoder"].captured.values()))
            # the decoder stack's differentiable inputs (token rows, [video | audio] rows) get their gradients from the captured backward
            assert [st.requires_grad for st in dec.static_in[:2]] == [True, True] and all(st.grad is not None for st in dec.static_in[:2])
            assert dec.draws > 0
            assert all(c.sunk for s in segs.values() for c in s.captured.values())          # gradient writes recorded for the reducer
            assert next(iter(segs["ast"].captured.values())).draws > 0 and next(iter(segs["vit"].captured.values())).draws == 0
        runs[graphs] = (losses, model.arena.flat.clone())
        model.enable_graphs(False)
        eng.close()
        del model, eng
        device_rng.disable_device_base()
    assert runs[False][0] == runs[True][0], (runs[False][0], runs[True][0])
    assert torch.equal(runs[False][1], runs[True][1])
    assert len({round(l["contra_loss"], 6) for l in runs[True][0]}) == 6                    # the steps do differ (training moves, masks change)


def test_graphed_videoswin_encoder_with_stochastic_depth(dev, device_rng):
    """the VideoSwin variant (scripts/pretrain.sh): the 3-D shifted-window encoder -- index-map gathers, window attention with its bias-table
    gradient, PatchMerging, stochastic depth 0.2 whose per-sample keep factors are drawn on the host and enter the graph as an input --
    and the AST encoder replayed, the shared-BERT text pass and the decoder eager: six steps bit-identical to six eager steps."""
    import numpy as np
    from valor_amd import ops
    runs = {}
    for graphs in (False, True):
        model, eng, batch = _engine(dev, graphs, swin=True)
        ops.DropoutState.reset(78)
        random.seed(6)
        np.random.seed(7)
        losses = []
        for step in range(6):
            out = eng.train_step(batch, TASK)
            losses.append({k: float(v) for k, v in out.items()})
        torch.cuda.synchronize()
        if graphs:
            assert set(model._graph_segs) == {"swin_droppath", "ast", "decoder"} and all(len(s.captured) == 1 for s in model._graph_segs.values())
        runs[graphs] = (losses, model.arena.flat.clone())
        model.enable_graphs(False)
        eng.close()
        del model, eng
        device_rng.disable_device_base()
    assert runs[False][0] == runs[True][0], (runs[False][0], runs[True][0])
    assert torch.equal(runs[False][1], runs[True][1])


def test_replays_draw_fresh_dropout_masks(dev, device_rng):
    """the AST encoder's captured graph (dropout 0.1 after every sub-layer and on the attention probabilities) replayed on the SAME input:
    another step -> another output; the same step counter -> the same output (the graph itself is deterministic)."""
    from valor_amd import ops
    model, eng, batch = _engine(dev, True)
    ops.DropoutState.reset(5)
    random.seed(1)
    for _ in range(3):
        eng.train_step(batch, TASK)
    seg = model._graph_segs["ast"]
    assert len(seg.captured) == 1
    model.train()
    aud = batch["audio_spectrograms"]
    outs = []
    for bump in (True, True, False):
        if bump:
            ops.DropoutState.begin_step()
        else:
            ops.DropoutState.offset = 0
        outs.append(model.forward_audio_encoder(aud).detach().clone())
    assert not torch.equal(outs[0], outs[1])
    assert torch.equal(outs[1], outs[2])
    model.enable_graphs(False)
    eng.close()


def test_reducer_sees_the_gradient_writes_of_a_replayed_backward(dev, device_rng):
    """the data-parallel bookkeeping is host code beside the launches: after a replay the recorded names are reported again, so the
    reducer's per-step picture (names touched, write counts) is the eager step's"""
    from valor_amd import ops
    pictures = {}
    for graphs in (False, True):
        model, eng, batch = _engine(dev, graphs)
        ops.DropoutState.reset(5)
        random.seed(1)
        for _ in range(4):
            eng.train_step(batch, TASK)
        pictures[graphs] = dict(eng.reducer.touched)
        model.enable_graphs(False)
        eng.close()
        del model, eng
        device_rng.disable_device_base()
    assert pictures[True] == pictures[False] and len(pictures[True]) > 50


def test_evaluation_between_training_steps_invalidates_the_decoder_graph(dev, device_rng):
    """model.eval() releases the static K|V buffers of the decoder's cross-attention (VALOR.train / release_static_kv) -- whose addresses a
    captured decoder stack has baked in. The captured graphs must go with them: training steps, a validation forward, more training steps
    equal the eager run of the same sequence to the bit (train_utils.py:366-372 validates in the middle of training)."""
    from valor_amd import ops
    runs = {}
    for graphs in (False, True):
        model, eng, batch = _engine(dev, graphs)
        ops.DropoutState.reset(79)
        random.seed(8)
        losses = []
        for step in range(4):
            losses.append({k: float(v) for k, v in eng.train_step(batch, TASK).items()})
        if graphs:
            assert len(model._graph_segs["decoder"].captured) == 1
        model.eval()
        with torch.no_grad():
            ev = model(batch, task=TASK, compute_loss=False)
        scores = ev["caption_scores_tva"].float().cpu().clone()
        if graphs:
            assert len(model._graph_segs["decoder"].captured) == 0          # gone with the buffers
        for step in range(4):                                               # two eager steps, the re-capture, a replay
            losses.append({k: float(v) for k, v in eng.train_step(batch, TASK).items()})
        torch.cuda.synchronize()
        if graphs:
            assert len(model._graph_segs["decoder"].captured) == 1
        runs[graphs] = (losses, model.arena.flat.clone(), scores)
        model.enable_graphs(False)
        eng.close()
        del model, eng
        device_rng.disable_device_base()
    assert runs[False][0] == runs[True][0]
    assert torch.equal(runs[False][1], runs[True][1]) and torch.equal(runs[False][2], runs[True][2])
