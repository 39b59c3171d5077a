"""SURVEY 8 row f4, "a working KV cache" (the reference's is disabled for a decoder with cross-attention and broken behind that,
model/pretrain.py:890-896, model/bert.py:807): valor_amd.decode.DecodeSession -- two rows per sequence and step against per-layer K|V
slots, the step captured as a hipGraph -- against the path that re-runs every text row each step like the reference (decode._Stepper),
which tests/test_finetune_gpu.py pins on the oracle and on the reference's golden sequences.

fp32: teacher-forced logits of every step within 2e-5, generated ids identical (greedy, beam-3, video QA with padded questions, a
generated id 0 = a masked key, rows that end). bf16 at base widths: the logits of the two paths differ by less than the bf16 logit band
of tests/test_finetune_gpu.py. Graph replay == eager cached step, bit for bit; a session reused for a second batch gives what a fresh
one gives."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def _build(spec, sd, dtype, dev, **opts):
    from valor_amd.model.valor import VALOR
    m = VALOR({"dropout": 0.0, "drop_path_rate": 0.0, **opts}, spec=spec, dtype=dtype, device=dev)
    m.load_state_dict(sd, strict=True)
    m.eval()
    return m


def _paths(model, batch, task, monkeypatch, **attrs):
    """the same generation call through the cached graph, the cached eager step and the re-run path"""
    from valor_amd import decode
    for k, v in attrs.items():
        setattr(model, k, v)
    outs = []
    for cache, graph in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("VALOR_KV_CACHE", cache)
        monkeypatch.setenv("VALOR_DECODE_GRAPH", graph)
        decode.release_sessions(model)
        with torch.no_grad():
            outs.append({k: v.cpu() for k, v in model(batch, task=task, compute_loss=False).items()})
    decode.release_sessions(model)
    return outs


@pytest.mark.parametrize("table", ["0", "1"])                     # beam search: slots gathered into a second buffer | a row table (decode.py)
@pytest.mark.parametrize("prompt", [False, True])
@pytest.mark.parametrize("caption_type", ["unimlm", "lm"])
def test_cached_generation_equals_the_rerun_path_fp32(dev, monkeypatch, prompt, caption_type, table):
    from valor_amd import synth
    monkeypatch.setenv("VALOR_BEAM_TABLE", table)
    spec = synth.tiny_spec()
    sd = synth.make_state_dict(spec, seed=5, w_std=0.05)
    batch = synth.make_batch(spec, batch=3, frames=2, audio_slices=2, txt_len=32, seed=6)
    model = _build(spec, sd, torch.float32, dev, use_task_prompt=prompt, caption_type=caption_type, max_generation_len=9)
    for beam in (1, 3):
        g, e, r = _paths(model, batch, "cap%tva%tv%ta", monkeypatch, beam_size=beam)
        for k in r:
            assert torch.equal(g[k], e[k]), (beam, k)                  # replay == eager cached step
            assert torch.equal(g[k], r[k]), (beam, k, g[k], r[k])      # cached == re-run
    # video QA: the questions are the prompt rows, zero-padded per clip (masked keys among the prompt slots)
    qb = synth.make_batch(spec, batch=3, frames=2, audio_slices=2, txt_len=10, seed=7, questions=True)
    assert (qb["question_tokens"]["bert_tokens"] == 0).any()
    for bq in (1, 3):
        g, e, r = _paths(model, qb, "qa%tva%tv", monkeypatch, beam_size_qa=bq, beam_size=3, max_generation_len=6)
        for k in r:
            assert torch.equal(g[k], e[k]) and torch.equal(g[k], r[k]), (bq, k, g[k], r[k])


def test_cached_step_logits_match_teacher_forced_rerun(dev, monkeypatch):
    """every step's logits, teacher-forced on a fixed sequence that contains the id 0 (a generated [PAD] is a masked key from then on,
    bert.py:857) and [SEP]; fp32 2e-5"""
    from valor_amd import decode, synth
    spec = synth.tiny_spec()
    sd = synth.make_state_dict(spec, seed=11, w_std=0.05)
    batch = synth.make_batch(spec, batch=4, frames=2, audio_slices=1, txt_len=32, seed=12)
    for prompt in (False, True):
        model = _build(spec, sd, torch.float32, dev, use_task_prompt=prompt)
        T = 7
        gen = torch.Generator().manual_seed(3)
        seq = torch.randint(1000, spec.vocab, (4, T), generator=gen)
        seq[1, 2] = 0
        seq[2, 0] = 0
        seq[3, 4] = decode.EOS
        with torch.no_grad():
            b, kv, ranges = decode.encode_for_generation(model, batch, ["tva"])
            ref = decode.stepper(model, "tva", b, kv, ranges)
            p = model.get_task_prompt(decode.PROMPTS["caption"], b) if prompt else None
            worst = 0.0
            for graph in ("0", "1"):
                monkeypatch.setenv("VALOR_DECODE_GRAPH", graph)
                decode.release_sessions(model)
                sess = decode.session(model, b, 1, 0 if p is None else p.shape[1], T + 1, kv)
                sess.begin_batch(kv)
                for rep in range(2):                                   # the second round replays the captured step on a reset session
                    sess.begin_group(ranges["tva"], p)
                    for t in range(T + 1):
                        lc = sess.step(None if t == 0 else seq[:, t - 1].to(dev)).clone()
                        lr = ref.logits(seq[:, :t] if t else None, b)
                        worst = max(worst, float((lc - lr).abs().max()))
                        assert torch.equal(lc.argmax(-1), lr.argmax(-1)), (graph, rep, t)
            assert worst < 2e-5, worst
        decode.release_sessions(model)


def test_session_is_reused_across_batches_and_ends_rows(dev, monkeypatch):
    """one session (one captured step) serves batch after batch of a geometry: the second batch's sequences equal a fresh session's; with the
    [SEP] bias raised some rows end early and stay [SEP]"""
    from valor_amd import decode, synth
    spec = synth.tiny_spec()
    sd = synth.make_state_dict(spec, seed=21, w_std=0.05)
    sd["cls.decoder.bias"] = sd["cls.decoder.bias"].clone()
    model = _build(spec, sd, torch.float32, dev, beam_size=1, max_generation_len=12)
    b1 = synth.make_batch(spec, batch=3, frames=2, audio_slices=2, txt_len=32, seed=22)
    b2 = synth.make_batch(spec, batch=3, frames=2, audio_slices=2, txt_len=32, seed=23)
    with torch.no_grad():
        # find a bias for [SEP] that ends some rows and not others
        monkeypatch.setenv("VALOR_KV_CACHE", "0")
        for delta in (0.0, 0.5, 1.0, 2.0, 4.0):
            sd2 = dict(sd)
            sd2["cls.decoder.bias"] = sd["cls.decoder.bias"].clone()
            sd2["cls.decoder.bias"][decode.EOS] += delta
            model.load_state_dict(sd2, strict=True)
            r2 = model(b2, task="cap%tva%tv", compute_loss=False)["generated_sequences_t_va"].cpu()
            ended = (r2 == decode.EOS).any(dim=1)
            if ended.any() and not (r2 == decode.EOS).all():
                break
        r1 = {k: v.cpu() for k, v in model(b1, task="cap%tva%tv", compute_loss=False).items()}
        r2 = {k: v.cpu() for k, v in model(b2, task="cap%tva%tv", compute_loss=False).items()}
        monkeypatch.setenv("VALOR_KV_CACHE", "1")
        decode.release_sessions(model)
        c1 = {k: v.cpu() for k, v in model(b1, task="cap%tva%tv", compute_loss=False).items()}
        sessions = dict(model._decode_sessions)
        c2 = {k: v.cpu() for k, v in model(b2, task="cap%tva%tv", compute_loss=False).items()}
        assert list(model._decode_sessions.values()) == list(sessions.values()) and len(sessions) == 1       # same session object, reused
        assert next(iter(sessions.values())).graphs
    for k in r1:
        assert torch.equal(c1[k], r1[k]), k
        assert torch.equal(c2[k], r2[k]), k
    decode.release_sessions(model)


@pytest.mark.parametrize("name", ["ref_base_b2f2a1_ft"])
def test_base_widths_bf16_cached_logits_within_the_band(dev, name):
    """bf16 at base widths, teacher-forced on the REFERENCE's greedy sequence: the cached step's logits stay within the bf16 logit band
    (tests/test_finetune_gpu.py) of the fp32 native logits and of the re-run path's bf16 logits; fp32 cached argmax == the reference token"""
    from test_model_gpu import _recipe_tensors
    from test_finetune_gpu import BF16_LOGIT_BAND, GOLD
    from valor_amd import decode
    g = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    spec, sd, batch = _recipe_tensors(g["recipe"])
    ref_seq = g["greedy"]["generated_sequences_t_va"]
    T = ref_seq.shape[1]
    logits = {}
    with torch.no_grad():
        for dt in (torch.bfloat16, torch.float32):
            m = _build(spec, sd, dt, dev)
            b, kv, ranges = decode.encode_for_generation(m, batch, ["tva"])
            sess = decode.session(m, b, 1, 0, T, kv)
            sess.begin_batch(kv)
            sess.begin_group(ranges["tva"], None)
            rr = decode.stepper(m, "tva", b, kv, ranges)
            logits[dt] = [(sess.step(None if t == 0 else ref_seq[:, t - 1].to(dev)).cpu(), rr.logits(ref_seq[:, :t] if t else None, b).cpu())
                          for t in range(T)]
            decode.release_sessions(m)
    worst32 = worst16 = 0.0
    for t in range(T):
        c16, r16 = logits[torch.bfloat16][t]
        c32, r32 = logits[torch.float32][t]
        assert torch.equal(c32.argmax(-1), ref_seq[:, t])
        worst32 = max(worst32, float((c32 - r32).abs().max()))
        worst16 = max(worst16, float((c16 - c32).abs().max()), float((c16 - r16).abs().max()))
    assert worst32 < 1e-4, worst32
    assert worst16 < BF16_LOGIT_BAND, worst16
    print(f"{name}: cached vs re-run logits fp32 {worst32:.2e}; bf16 cached vs fp32 / vs bf16 re-run {worst16:.4f} over {T} teacher-forced steps")


@pytest.mark.parametrize("b,cur,beam,V,beam_major", [(64, 3, 3, 30522, True), (5, 1, 3, 30522, True), (7, 4, 4, 1000, False), (3, 8, 8, 2500, True), (2, 2, 1, 17, False)])
def test_beam_select_kernel_matches_the_torch_arithmetic(dev, b, cur, beam, V, beam_major):
    """valor_beam_select against the reference's arithmetic (pretrain.py:1080-1098: log-softmax subtract, add, the 0 / 1 mask products) and
    torch.topk: identical VALUES (bits) and beams everywhere, identical words wherever the chosen beam is open; a beam that has ended
    supplies V equal candidates (the reference's unstable sort leaves their order open): the kernel takes them in index order."""
    from valor_amd import decode
    g = torch.Generator().manual_seed(b * 100 + cur * 10 + beam)
    logits = (torch.randn((b * cur, V), generator=g) * 3).to(dev)
    seq_lp = (-torch.rand((b, cur, 1), generator=g) * 5).to(dev)
    seq_mask = (torch.rand((b, cur, 1), generator=g) > 0.3).float().to(dev)
    if cur > 1:
        seq_lp[0, 1] = 1.0                               # an ended beam with the best score of its sample: fills the whole selection
        seq_mask[0, 1] = 0.0
    lse = torch.empty(b * cur, device=dev)
    val, idx = decode.beam_select(logits, b, cur, beam, seq_lp, seq_mask, beam_major, lse_out=lse)      # the rows' log-sum-exp inside the launch
    assert (lse.double() - torch.logsumexp(logits.double(), 1)).abs().max().item() < 1e-5

    def torch_select(wl):
        wl = wl.view(cur, b, V).transpose(0, 1) if beam_major else wl.view(b, cur, V)
        cand = seq_lp + wl
        cand = seq_mask * cand + seq_lp.expand_as(cand) * (1 - seq_mask)
        return torch.topk(cand.reshape(b, -1), beam, dim=-1, largest=True, sorted=True)
    tv, ti = torch_select(logits - lse[:, None])
    assert torch.equal(val, tv)
    vx, ix = decode.beam_select(logits, b, cur, beam, seq_lp, seq_mask, beam_major, lse="xent")          # ... or handed in (the cross-entropy kernel's)
    tvx, tix = torch_select(decode.log_softmax_rows(logits))
    assert torch.equal(vx, tvx) and torch.equal(ix // V, tix // V)
    kb, tb = idx // V, ti // V
    assert torch.equal(kb, tb)
    open_ = torch.gather(seq_mask.view(b, cur), 1, kb) != 0
    assert torch.equal(idx[open_], ti[open_])
    if cur > 1:
        assert idx[0].tolist() == [V + i for i in range(beam)]
    # rows of the session's zero-padded buffer are read in place
    Vpad = (V + 31) // 32 * 32
    padded = torch.zeros((b * cur, Vpad), device=dev)[:, :V]
    padded.copy_(logits)
    v2, i2 = decode.beam_select(padded, b, cur, beam, seq_lp, seq_mask, beam_major)
    assert torch.equal(v2, val) and torch.equal(i2, idx)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,J,P,L,t", [(6, 2, 0, 10, 0), (5, 2, 3, 14, 7), (4, 1, 2, 9, 6), (300, 2, 5, 37, 30)])
def test_decode_prologue_kernel_equals_the_torch_ops(dev, dtype, R, J, P, L, t):
    """valor_decode_prologue (embeddings before their LayerNorm, the key-validity slot, the step's mask rows, its slots) against the torch
    expressions of DecodeSession._body it replaces: identical bits."""
    from valor_amd import kernels as K, lib
    from valor_amd.decode import MASK, NEG
    E, V = 192, 500
    g = torch.Generator().manual_seed(R + L + t)
    word, pos, typ = [torch.randn(s, generator=g).to(dev, dtype) for s in ((V, E), (64, E), (2, E))]
    tok = torch.randint(0, V, (R,), generator=g).to(dev)
    tok[::3] = 0                                                 # generated id 0: the key stays masked
    kmask = torch.where(torch.rand((R, L), generator=g) > 0.5, 0.0, NEG).to(dev)
    tdev = torch.tensor([t], device=dev)
    km, amask = kmask.clone(), torch.full((R, J, L), 7.0, device=dev)
    x, slots = torch.empty((R, J, E), dtype=dtype, device=dev), torch.zeros(J, dtype=torch.int64, device=dev)
    lib.call("valor_decode_prologue", torch.cuda.current_stream().cuda_stream, K.dt_of(x), tok.data_ptr(), tdev.data_ptr(), word.data_ptr(),
             pos.data_ptr(), typ.data_ptr(), MASK, R, J, E, P, L, NEG, km.data_ptr(), amask.data_ptr(), x.data_ptr(), slots.data_ptr())
    ids = torch.full((R, J), MASK, dtype=torch.int64, device=dev)
    ids[:, 0] = tok
    jidx, slot = torch.arange(J, device=dev), torch.arange(L, device=dev)
    want_x = (word[ids].float() + pos[tdev + jidx].float()[None] + typ[0].float()).to(dtype)
    slot_t = tdev + P
    want_km = kmask.clone()
    want_km.index_copy_(1, slot_t, torch.where(tok != 0, 0.0, NEG)[:, None])
    row_a = torch.where(slot <= slot_t, want_km, NEG)
    assert torch.equal(x, want_x) and torch.equal(km, want_km)
    assert torch.equal(amask[:, 0], row_a)
    if J == 2:
        assert torch.equal(amask[:, 1], torch.where(slot == slot_t + 1, 0.0, row_a))
    assert slots.tolist() == [P + t + j for j in range(J)]
