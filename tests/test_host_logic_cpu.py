"""CPU: host-side logic of the native model (no kernels): parameter table <-> reference checkpoint layout, optimizer
param groups, LR schedule, TokenMasker draw order, attention masks, task prompts, arena layout, FLOP model."""
import os
import random
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from valor_amd import synth  # noqa: E402
from valor_amd.model.params import optimizer_group, param_table  # noqa: E402
from valor_amd.model.valor import VALOR, TokenMasker  # noqa: E402
import valor_oracle as VO  # noqa: E402


def test_param_table_covers_reference_layout_exactly():
    spec = synth.base_spec()
    layout = synth.state_dict_layout(spec)
    ref_keys = {k: s for k, s, _ in layout}
    seen = {}
    n_params = 0
    for name, shape, refs in param_table(spec):
        numel = 1
        for s in shape:
            numel *= s
        n_params += numel
        if len(refs) == 1 or refs[1] == "cls.decoder.weight":
            for r in refs:
                seen[r] = shape
        else:
            rows = shape[0] // len(refs)
            for r in refs:
                seen[r] = (rows,) + tuple(shape[1:])
    assert set(seen) == set(ref_keys)
    for k, s in seen.items():
        assert tuple(s) == tuple(ref_keys[k]), (k, s, ref_keys[k])
    assert len(ref_keys) == 845 and n_params == 374680383      # 845 tensors, 374.68 M parameters


def test_state_dict_roundtrip_and_packing():
    spec = synth.tiny_spec()
    sd = synth.make_state_dict(spec, seed=1)
    m = VALOR(None, spec=spec, dtype=torch.float32, device="cpu")
    assert m.load_state_dict(sd, strict=True) == ([], [])
    out = m.state_dict()
    assert set(out) == set(sd) and all(torch.equal(out[k], sd[k]) for k in sd)
    p = "multimodal_encoder.encoder.layer.1.attention.self."
    packed = m.P[p + "qkv.weight"]
    H = spec.hidden
    assert torch.equal(packed[H:2 * H], sd[p + "key.weight"])          # fused-QKV view is the checkpoint's q|k|v
    assert out["cls.decoder.weight"].data_ptr() == out["multimodal_encoder.embeddings.word_embeddings.weight"].data_ptr()
    # arena: every tensor starts on an optimizer chunk, grads alias the flat gradient buffer
    for name, (off, n, shape) in m.arena.offsets.items():
        assert off % m.arena.chunk == 0
        assert m.P[name].grad.data_ptr() == m.arena.grad[off:].data_ptr()


def test_optimizer_groups_match_reference_rules():
    """optim/misc.py:13-64, including its case-sensitive quirks."""
    g = optimizer_group
    assert g("multimodal_encoder.encoder.layer.0.attention.self.query.weight") == 0
    assert g("multimodal_encoder.encoder.layer.0.attention.output.LayerNorm.weight") == 1
    assert g("audio_encoder.layer.0.layernorm1.weight") == 0            # lower-case 'layernorm' IS decayed
    assert g("cls.layernorm.weight") == 0 and g("cls.layernorm.bias") == 1
    assert g("clip_model.visual.ln_pre.weight") == 4 and g("clip_model.visual.ln_pre.bias") == 5
    assert g("clip_model.transformer.resblocks.0.attn.in_proj_bias") == 7 and g("clip_model.logit_scale") == 6
    assert g("contra_head_a.linear.weight", ("contra_head",)) == 2
    for k, _, _ in synth.state_dict_layout(synth.base_spec()):
        assert g(k) == VO.param_group_of(k), k


def test_lr_schedule():
    from valor_amd.optim import warmup_linear
    for x in (0.0, 0.05, 0.1, 0.5, 1.0, 1.2):
        assert warmup_linear(x, 0.1) == VO.warmup_linear(x, 0.1)
    assert warmup_linear(0.05, 0.1) == 0.5 and warmup_linear(1.0, 0.1) == 0.0


def test_token_masker_matches_reference_draw_order():
    spec = synth.base_spec()
    batch = synth.make_batch(spec, batch=5, frames=1, audio_slices=1, seed=3)
    toks = batch["txt_tokens"]["bert_tokens"]
    orc = VO.Oracle(spec, {}, vocab_tokens=synth.synthetic_vocab(spec.vocab))
    mk = TokenMasker(103, 106, spec.vocab)
    for prob in (0.6, 0.15):
        random.seed(42); a = orc.text_masker(toks, prob)
        random.seed(42); b = mk(toks, prob)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert ((b[1] != -1).sum(1) >= 1).all() and (b[1][:, 0] == -1).all()


def test_masks_and_prompts():
    spec = synth.tiny_spec()
    m = VALOR(None, spec=spec, dtype=torch.float32, device="cpu")
    toks = torch.tensor([[101, 5, 6, 102, 0, 0], [101, 7, 102, 0, 0, 0]])
    prompt = m.get_task_prompt("predict masked tokens with visual and audio cues", 2)
    assert prompt.shape == (2, 10) and prompt[0, 0] == 101 and prompt[0, -1] == 102 and (prompt != 100).all()
    am = m._bert_mask(toks, prompt, casual=True)
    assert am.shape == (2, 16, 16)
    assert am[0, 1, 2] == -10000.0 and am[0, 2, 1] == 0.0          # causal inside the text block
    assert (am[0, 6:, :6] == -10000.0).all()                       # prompt rows cannot see text (bert.py:882)
    assert am[0, 0, 4] == -10000.0 and am[0, 0, 6] == 0.0          # padding masked, prompt visible
    am2 = m._bert_mask(toks, prompt, casual=False)
    assert am2[0, 1, 2] == 0.0 and am2[0, 7, 1] == 0.0 and am2[0, 1, 5] == -10000.0
    cm = m._clip_text_mask(toks)
    assert cm[1, 2, 1] == 0.0 and cm[1, 1, 2] == -10000.0 and cm[1, 4, 3] == -10000.0


def test_unsupported_configs_fail_loudly():
    import pytest
    with pytest.raises(NotImplementedError):       # opts and spec disagree
        VALOR({"video_encoder_type": "videoswin_base_k600_22k"}, spec=synth.tiny_spec(), dtype=torch.float32, device="cpu")
    with pytest.raises(NotImplementedError):       # not a shipped combination (the reference loads CLIP as a whole)
        VALOR({"video_encoder_type": "videoswin_small_k400_1k"}, spec=synth.tiny_swin_spec(), dtype=torch.float32, device="cpu")
    with pytest.raises(NotImplementedError):
        VALOR({"video_encoder_type": "clip_vit_large_14_336px"}, spec=synth.tiny_spec(), dtype=torch.float32, device="cpu")
    with pytest.raises(NotImplementedError):
        VALOR({"fineweight_type": "none"}, spec=synth.tiny_spec(), dtype=torch.float32, device="cpu")
    m = VALOR(None, spec=synth.tiny_spec(), dtype=torch.float32, device="cpu")
    with pytest.raises(NotImplementedError):
        m({}, task="scst%tv")    # pt_ / ret% / cap% / qa% are the task families of VALOR.forward (tests/test_finetune_gpu.py)


def test_param_tables_of_the_swin_and_large_configurations():
    """every variant's internal parameter table covers its reference-keyed layout exactly (scripts/pretrain.sh VideoSwin-B:
    314.40 M parameters as probed from the reference; BASELINE configs[3] widths: both hidden_trans projections present)"""
    import math
    for spec, n_expected in ((synth.swin_spec(), 314404358), (synth.large_spec(), None), (synth.tiny_large_spec(), None), (synth.tiny_swin_spec(), None)):
        T = param_table(spec)
        refs = [r for _, _, rs in T for r in rs]
        lay = [k for k, _, kind in synth.state_dict_layout(spec) if kind not in ("alias", "relidx")]
        assert sorted(refs) == sorted(lay)
        if n_expected:
            assert sum(math.prod(s) for _, s, _ in T) == n_expected
    names = {n for n, _, _ in param_table(synth.large_spec())}
    assert "hidden_trans_audio_multimodal.0.weight" in names and "hidden_trans_video_multimodal.0.weight" in names
    assert dict((n, s) for n, s, _ in param_table(synth.large_spec()))["video_encoder.layers.2.downsample.norm.weight"] == (3072,)


def test_swin_variant_host_side():
    """VideoSwin + BERT-text variant (scripts/pretrain.sh:3-8): reference-keyed state dict round trip (integer buffers and the
    txt_encoder.* aliases included), window index maps == roll + window_partition / compute_mask of the oracle, PatchMerging
    gather rows, optimizer groups of the new tensors."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from valor_oracle import Oracle
    from valor_amd.model.params import optimizer_group
    spec = synth.tiny_swin_spec()
    m = VALOR({"video_encoder_type": "videoswin_base_k600_22k", "txt_encoder_type": "bert_base_uncased", "dropout": 0.0}, spec=spec,
              dtype=torch.float32, device="cpu")
    sd = synth.make_state_dict(spec, seed=3)
    assert m.load_state_dict(sd, strict=True) == ([], [])
    out = m.state_dict()
    assert set(out) == set(sd)
    for k in sd:
        assert torch.equal(out[k].cpu(), sd[k]), k
    assert out["txt_encoder.embeddings.word_embeddings.weight"].data_ptr() == out["multimodal_encoder.embeddings.word_embeddings.weight"].data_ptr()
    for size, shifted in (((16, 14, 14), True), ((8, 14, 14), True), ((8, 14, 14), False), ((2, 28, 28), True), ((8, 7, 7), True)):
        D, H, W = size
        geo = m._swin_geometry(D, H, W, shifted)
        win, sh = Oracle.swin_effective_window(size, spec.swin_window, tuple(v // 2 for v in spec.swin_window) if shifted else (0, 0, 0))
        ids = torch.arange(D * H * W, dtype=torch.float32).reshape(1, D, H, W, 1)
        if any(sh):
            ids = torch.roll(ids, shifts=(-sh[0], -sh[1], -sh[2]), dims=(1, 2, 3))
        assert torch.equal(Oracle.swin_windows(ids, win).reshape(-1).long(), geo["rowmap"].long())
        assert geo["N"] == win[0] * win[1] * win[2] and geo["nW"] * geo["N"] == D * H * W
        if any(sh):
            lab = geo["label"].long().view(geo["nW"], geo["N"])
            assert torch.equal(torch.where(lab[:, None, :] != lab[:, :, None], -100.0, 0.0), Oracle.swin_shift_mask(size, win, sh))
        else:
            assert geo["label"] is None
        full = synth.swin_relative_position_index(spec.swin_window)[:geo["N"], :geo["N"]]
        rel = geo["rel"].long()
        assert torch.equal(rel[:, None] - rel[None, :] + geo["relc"], full)
    assert m._swin_geometry(8, 16, 16, False)["padded"]        # 16 -> 21: zero padding, test_swin_padded_geometry_maps
    x = torch.arange(2 * 2 * 4 * 4).float().view(2, 2, 4, 4, 1)
    ref = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1).reshape(-1)
    assert torch.equal(ref.long(), m._swin_merge_idx(2, 2, 4, 4))
    # optim/misc.py:14: the bias TABLE matches "bias" (no decay), Swin norm weights are decayed, everything is in the basic family
    assert optimizer_group("video_encoder.layers.0.blocks.0.attn.relative_position_bias_table") == 1
    assert optimizer_group("video_encoder.layers.0.blocks.0.norm1.weight") == 0
    assert optimizer_group("hidden_trans_video_multimodal.1.weight") == 0 and optimizer_group("contra_head_v.linear.weight") == 0


def test_flop_model_matches_survey():
    sys.path.insert(0, ROOT)
    import bench
    nf = bench.necessary_flops_per_sample(synth.base_spec(), 8, 2, 32)
    assert 1.15e12 < nf < 1.30e12          # ~1218 GFLOP fwd+bwd per sample with shared cross-K/V


def test_attention_masks_match_reference_construction():
    """numpy mask builders == the torch construction of bert.py:854-885 / clip.py:382-414"""
    import torch
    from valor_amd.model.valor import VALOR

    def ref_bert(tokens, prompt, casual):
        am = (tokens != 0).long()
        token_len = am.shape[1]
        if prompt is not None:
            am = torch.cat((am, (prompt != 0).long()), dim=1)
        total = am.shape[1]
        am = am.unsqueeze(1).expand(-1, total, -1).clone()
        if casual:
            am[:, :token_len, :token_len] = torch.tril(am[:, :token_len, :token_len])
            am[:, token_len:, :token_len] = 0
        return (1.0 - am.float()) * -10000.0

    g = torch.Generator().manual_seed(0)
    tok = torch.randint(0, 4, (5, 32), generator=g)
    pr = torch.randint(0, 3, (5, 10), generator=g)
    for casual in (True, False):
        for p in (None, pr):
            assert torch.equal(ref_bert(tok, p, casual), VALOR._bert_mask(tok, p, casual))
    L = tok.shape[1]
    m = torch.tril((tok != 0).long().unsqueeze(1).expand(-1, L, -1).clone())
    assert torch.equal((1.0 - m.float()) * -10000.0, VALOR._clip_text_mask(tok))


def test_host_stage_passthrough_on_cpu():
    """without a GPU the staging ring is a no-op that still honours dtype conversion"""
    import torch
    from valor_amd.hoststage import HostStage
    st = HostStage("cpu")
    st.begin_step()
    t = torch.arange(6).view(2, 3)
    assert torch.equal(st.put(t), t) and st.put(t, torch.float32).dtype == torch.float32


def test_swin_padded_geometry_maps():
    """VideoSwin window padding lives in index maps (valor.py _swin_geometry / _swin_pad_idx): pad -> roll -> window_partition of the
    reference (videoswin.py:198-211) and compute_mask on the padded map (:335-338), against the oracle's helpers; PatchMerging's zero
    padding of odd maps (:257-259) against F.pad + strided slicing."""
    import sys, os
    import torch.nn.functional as F
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from valor_oracle import Oracle
    from valor_amd import synth
    from valor_amd.model.valor import VALOR
    m = VALOR({"dropout": 0.0, "drop_path_rate": 0.0}, spec=synth.tiny_swin_spec(), dtype=torch.float32, device="cpu")
    for size, shifted in (((10, 28, 28), True), ((10, 28, 28), False), ((3, 26, 26), True), ((3, 13, 13), True), ((10, 13, 13), True),
                          ((3, 6, 6), True), ((8, 14, 14), True)):
        D, H, W = size
        geo = m._swin_geometry(D, H, W, shifted)
        win, sh = Oracle.swin_effective_window(size, m.spec.swin_window, tuple(v // 2 for v in m.spec.swin_window) if shifted else (0, 0, 0))
        Dp, Hp, Wp = [-(-s // w) * w for s, w in zip(size, win)]
        assert geo["padded"] == ((Dp, Hp, Wp) != size)
        ids = torch.arange(D * H * W, dtype=torch.float32).reshape(1, D, H, W, 1) + 1            # 0 = a padding position
        idp = F.pad(ids, (0, 0, 0, Wp - W, 0, Hp - H, 0, Dp - D))
        if any(sh):
            idp = torch.roll(idp, shifts=(-sh[0], -sh[1], -sh[2]), dims=(1, 2, 3))
        want = Oracle.swin_windows(idp, win).reshape(-1).long() - 1                               # natural row, -1 = zero row
        rowmap = geo["rowmap"].long()
        if geo["padded"]:
            pad, unpad = m._swin_pad_idx(geo, 2)
            n, npad = D * H * W, Dp * Hp * Wp
            assert torch.equal(pad[:npad][rowmap], want)
            assert torch.equal(pad[npad:][rowmap], torch.where(want >= 0, want + n, want))        # second clip of the batch
            assert torch.equal(pad[unpad], torch.arange(2 * n))                                   # the crop x[:, :D, :H, :W]
        else:
            assert torch.equal(rowmap, want)
        if any(sh):
            lab = geo["label"].long().view(geo["nW"], geo["N"])
            assert torch.equal(torch.where(lab[:, None, :] != lab[:, :, None], -100.0, 0.0), Oracle.swin_shift_mask((Dp, Hp, Wp), win, sh))
        else:
            assert geo["label"] is None
    for (b, D, H, W) in ((2, 3, 13, 13), (1, 2, 7, 6), (2, 2, 4, 4)):
        x = torch.randn(b, D, H, W, 5)
        xp = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        want = torch.cat([xp[:, :, 0::2, 0::2], xp[:, :, 1::2, 0::2], xp[:, :, 0::2, 1::2], xp[:, :, 1::2, 1::2]], -1)
        idx = m._swin_merge_idx(b, D, H, W)
        rows = torch.cat((x.reshape(-1, 5), torch.zeros(1, 5)))[idx]                              # index -1 -> the appended zero row
        assert torch.equal(rows.reshape(want.shape), want)


@pytest.mark.skipif(not os.path.isdir("/root/reference/config"), reason="/root/reference not present")
def test_every_shipped_config_file_parses_and_names_a_built_task_family():
    """config/*.json of the reference (pretraining, fast-retrieval-*, caption-*, VQA-*): load_config accepts each file as shipped
    (two lack their closing brace), train_tasks lists its task strings, and every one belongs to a family VALOR.forward dispatches
    (pt_ / ret% / cap% / qa%, model/pretrain.py:125-135) with groups this build runs (tva / tv / ta)."""
    import glob
    from valor_amd.config import load_config, train_tasks
    files = sorted(glob.glob("/root/reference/config/*.json"))
    assert len(files) >= 20
    seen = set()
    for f in files:
        opts = load_config(f)
        tasks = train_tasks(opts)
        assert tasks, f
        for task, bs in tasks:
            fam = task.split("_")[0] if task.startswith("pt") else task.split("%")[0]
            assert fam in ("pt", "ret", "cap", "qa"), (f, task)
            seen.add(fam)
            groups = [g for part in (task.split("_")[1:] if fam == "pt" else [task]) for g in part.split("%")[1:]]
            assert groups and all(g in ("tva", "tv", "ta") for g in groups), (f, task)
        assert opts.video_resolution == 224 and opts.beam_size == 3 and opts.beam_size_qa == 1        # argparse defaults carried through
    assert seen == {"pt", "ret", "cap", "qa"}


def test_qa_prompt_splices_the_task_prompt_behind_cls():
    """model/pretrain.py:1268-1274: question rows as the prompt; with use_task_prompt 'answer the question' goes between the question's
    [CLS] and its first word -- native helper == oracle helper"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from valor_oracle import Oracle
    from valor_amd import synth
    from valor_amd.model.valor import VALOR
    spec = synth.tiny_spec()
    q = synth.make_batch(spec, batch=3, frames=1, audio_slices=1, txt_len=8, seed=2, questions=True)["question_tokens"]["bert_tokens"]
    for prompt in (False, True):
        m = VALOR({"dropout": 0.0, "use_task_prompt": prompt}, spec=spec, dtype=torch.float32, device="cpu")
        o = Oracle(spec, {}, vocab_tokens=synth.synthetic_vocab(spec.vocab), use_task_prompt=prompt)
        got, want = m.qa_prompt(q), o.qa_prompt(q)
        assert torch.equal(got, want)
        if prompt:
            assert got.shape[1] == q.shape[1] + 3 and torch.equal(got[:, 0], q[:, 0]) and torch.equal(got[:, 4:], q[:, 1:])
            assert (got[:, 1:4] != 100).all()                      # 'answer', 'the', 'question' are vocabulary words, not [UNK]
        else:
            assert got is q


def test_bench_refuses_a_rank_count_that_is_not_gpus():
    """bench.py --gpus N under a launcher that started a different number of ranks must fail loudly, not print n_gpus of its own
    (round-2 review: the flag was parsed and ignored)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)


def test_param_view_carries_the_gradient_slot_and_logits_rows_have_slack():
    """ops.param_view: a reshaped view of an arena parameter (the conv kernels used as GEMM weights, clip.py:236 / modeling.py:752)
    keeps the parameter's arena name and aliases the SAME reshape of its gradient slot, so the wgrad GEMM accumulates in place; a
    tensor that is not an arena parameter passes through as a plain view. ops._rows_with_slack: the [rows, vocabulary] logits of the
    masked rows are a view of a buffer with >= 12 % spare rows (the row count changes per step)."""
    from valor_amd import ops
    p = torch.nn.Parameter(torch.arange(24, dtype=torch.float32).view(2, 3, 2, 2))
    p._arena_name = "conv.weight"
    p.grad = torch.zeros_like(p)
    v = ops.param_view(p, 2, -1)
    assert v.shape == (2, 12) and v._arena_name == "conv.weight"
    assert v._sink_view.shape == (2, 12) and v._sink_view.data_ptr() == p.grad.data_ptr()
    assert ops._sink(v) is v._sink_view and ops._sink(p) is p.grad
    v._sink_view[1, 11] = 7.0
    assert float(p.grad[1, 2, 1, 1]) == 7.0
    q = torch.nn.Parameter(torch.zeros(4, 6))          # no arena name / no gradient slot: an ordinary view, gradients through autograd
    w = ops.param_view(q, 2, 12)
    assert getattr(w, "_sink_view", None) is None and ops._sink(w) is None
    sizes = []
    for n in (1, 290, 2100, 1153, 4096, 300):
        b = ops._rows_with_slack(n, 64, torch.bfloat16, "cpu")
        assert b.shape == (n, 64) and b.is_contiguous()
        assert b.untyped_storage().nbytes() >= int(n * 1.12) * 64 * 2 and b.untyped_storage().nbytes() % (512 * 64 * 2) == 0
        sizes.append(b.untyped_storage().nbytes())
    assert sizes == sorted(sizes) and sizes[2] == sizes[3] and sizes[4] == sizes[5]      # monotone: a smaller draw re-requests the largest size


def test_checkpoint_rewinds_the_dropout_windows():
    """ops.CheckpointFn (the reference's `checkpointing`, clip.py:208-209 / transformer.py:163-164): the layer runs again in backward and
    must draw the SAME dropout windows as its first run, while everything issued between the two runs keeps its own. Host logic only:
    a stand-in layer whose mask is derived from the (seed, offset) pair ops.DropoutState hands out."""
    import torch
    from valor_amd import ops

    drawn = []

    def layer(x, y):
        seed, off = ops.DropoutState.draw(x.numel())
        drawn.append(off)
        g = torch.Generator().manual_seed(seed * 1000003 + off)
        mask = (torch.rand(x.shape, generator=g) > 0.3).float()
        return x * mask * w + y, (y * 2.0)

    w = torch.nn.Parameter(torch.tensor(1.5))
    res = []
    for ck in (False, True):
        ops.DropoutState.reset(7)
        drawn.clear()
        w.grad = None
        x = torch.arange(12.0).reshape(3, 4).requires_grad_()
        y = torch.ones(3, 4, requires_grad=True)
        a, b = ops.checkpoint(layer, x, y) if ck else layer(x, y)
        seed2, later = ops.DropoutState.draw(5)                      # another op's window between forward and backward
        (a.sum() + 3.0 * b.sum()).backward()
        assert ops.DropoutState.offset == later + (5 + 3) // 4 + 1       # the counter is where the forward left it
        res.append((a.detach().clone(), x.grad.clone(), y.grad.clone(), w.grad.clone(), list(drawn)))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert torch.equal(res[0][3], res[1][3])
    assert res[0][4] == [0] and res[1][4] == [0, 0]                   # the second run of the checkpointed layer drew offset 0 again


def test_dropout_state_device_mode_bookkeeping():
    """ops.DropoutState: host mode hands out ever-growing by-value windows; device mode (enable_device_base) restarts them at every
    begin_step and advances the per-step counter the kernels add (here a CPU tensor stands in for the device word): the by-value
    numbers of a step are the same step after step -- what a captured graph bakes in -- while (counter + offset) never repeats."""
    import torch
    from valor_amd import kernels as K, ops
    D = ops.DropoutState
    D.disable_device_base()
    D.reset(5)
    a = [D.draw(1000), D.draw_elems(77), D.draw(4)]
    assert [o for _, o in a] == [0, 251, 251 + 78] and all(s == 5 for s, _ in a) and K.RNG_BASE == 0
    base = D.enable_device_base("cpu")
    try:
        assert K.RNG_BASE == base.data_ptr() and int(base) == 0 and D.offset == 0
        seen = set()
        for step in range(3):
            D.begin_step()
            assert int(base) == (step + 1) * D.STEP_STRIDE and D.offset == 0
            offs = [D.draw(1000)[1], D.draw_elems(77)[1]]
            assert offs == [0, 251]                                   # identical by-value windows every step
            eff = [int(base) + o for o in offs]
            assert not (seen & set(eff))
            seen |= set(eff)
        D.reset(9)
        assert int(base) == 0 and D.seed == 9
    finally:
        D.disable_device_base()
        D.reset(1234)
    assert K.RNG_BASE == 0 and D.base is None


def test_gradient_write_reports_are_recorded_during_capture_and_replayed():
    """valor_amd/graphs.py's capture-time schedule, host side: while ops.GradSink.recorder is set (a backward is being CAPTURED, nothing
    runs) neither the kernels' reports (ops._sunk) nor autograd's post-accumulate hooks reach the data-parallel reducer -- they are recorded
    in order; reporting the recorded names afterwards gives the reducer exactly the picture of an eager backward."""
    import torch
    from valor_amd import ops
    from valor_amd.arena import ParamArena
    from valor_amd.dist import Reducer
    arena = ParamArena([(f"p{i}", (64,), 0) for i in range(4)], torch.float32, "cpu")
    red = Reducer(arena, bucket_bytes=256)
    try:
        params = list(arena.params.values())

        def backward_of_a_step():
            (params[0] * 2.0 + params[1] * 3.0).sum().backward()      # autograd hooks report p0, p1
            ops._sunk(params[2])                                      # a kernel that wrote p2's gradient straight into the arena

        red.prepare_backward()
        backward_of_a_step()
        eager = dict(red.touched)
        assert eager == {"p0": 1, "p1": 1, "p2": 1}
        red.prepare_backward()
        rec = []
        ops.GradSink.recorder = rec
        try:
            backward_of_a_step()                                      # "capture": nothing may reach the reducer
        finally:
            ops.GradSink.recorder = None
        # kernel reports are recorded as names, hook reports as ("hook", name): a parameter that eager code of the same step uses too is
        # hooked by the OUTER backward as well, and the replay must then not report the captured hook a second time
        assert red.touched == {} and sorted(map(str, rec)) == sorted(map(str, [("hook", "p0"), ("hook", "p1"), "p2"]))
        outer_hooked = set()                                          # nothing outside the "graph" uses these parameters
        for name in rec:                                              # what graphs._Replay.backward does after every replay
            if isinstance(name, tuple):
                if name[1] not in outer_hooked:
                    ops.GradSink.listener(name[1])
            else:
                ops.GradSink.listener(name)
        assert red.touched == eager
        # the same with p0 ALSO used by eager code of the step (the shared-BERT text pass beside the graphed decoder): autograd hooks p0 once
        # per backward pass, so the eager count stays 1 + 1 + ... and the replay leaves the hook report of p0 to the outer backward
        red.prepare_backward()
        (params[0] * 5.0).sum().backward()                            # the outer backward's own use of p0: its hook runs live
        assert "p0" in ops.GradSink.live_hooks
        outer_hooked = {n[1] for n in rec if isinstance(n, tuple) and n[1] in ops.GradSink.live_hooks}
        for name in rec:
            if isinstance(name, tuple):
                if name[1] not in outer_hooked:
                    ops.GradSink.listener(name[1])
            else:
                ops.GradSink.listener(name)
        assert red.touched == {"p0": 1, "p1": 1, "p2": 1}
    finally:
        ops.GradSink.listener = None
        ops.GradSink.recorder = None
