"""Caption generation on the native decoder: VALOR.generate_cap / decode_greedy / decode_beam / get_logits
(model/pretrain.py:914-1189), caption_type 'unimlm'.

The reference feeds [CLS] + generated tokens + [MASK] through the whole decoder again at every step (its cache path is disabled when the
decoder has cross-attention, pretrain.py:890-896, bert.py:848-850). Here the video/audio K|V projections of the 12 layers are computed
ONCE per clip (VALOR.cross_inputs) and shared by every step, every group and every beam; the text rows are re-run each step like the
reference (<= 32 rows x (t + 2) tokens: launch-bound, not FLOP-bound).

Beam rows are kept in the reference's (sample, beam) order on the host side; the decoder sees them beam-major (row = beam * b + sample)
so that a row's cross-attention K|V is row % b -- the kv_bmod addressing the training passes use -- instead of beam copies of K|V
(pretrain.py:1133-1139 expands video_input / audio_input beam_size times).

Everything here is inference: torch.no_grad, dropout off regardless of model.training (the reference calls it under model.eval())."""
import torch

from . import kernels as K, lib, ops
from .model.valor import PROMPTS

BOS, EOS, MASK = 101, 102, 103          # [CLS] / [SEP] / [MASK] of bert-base-uncased, model/modeling.py:669-671


def _st():
    return K._stream()


class _Stepper:
    """get_logits + forward_cap_single(compute_loss=False), pretrain.py:1031-1051,882-900, for one query group."""

    def __init__(self, model, group, kv_layers, ranges, prompt_cpu, b):
        self.m, self.kv, self.b, self.prompt = model, kv_layers, b, prompt_cpu
        self.range = list(ranges[group]) if kv_layers is not None else None

    def logits(self, state, rows):
        """state: host int64 [rows, t] in (sample, beam) order (None at t = 0) -> fp32 logits [rows, vocab] of the last text position:
        the appended [MASK] (caption_type 'unimlm') or the last token itself ('lm': [CLS] + the tokens so far, pretrain.py:1038-1040).
        full_masker does not reach this path in the reference either (forward_cap_single's generation branch, :878-900, drops the flag)."""
        m, b = self.m, self.b
        beam = rows // b
        bos = torch.full((rows, 1), BOS, dtype=torch.long)
        if m.caption_type == "lm":
            txt = bos if state is None else torch.cat((bos, state), dim=1)
        else:
            mask_col = torch.full((rows, 1), MASK, dtype=torch.long)
            txt = torch.cat((bos, mask_col), dim=1) if state is None else torch.cat((bos, state, mask_col), dim=1)
        if beam > 1:                                                # (sample, beam) -> beam-major
            perm = torch.arange(rows).view(b, beam).t().reshape(-1)
            txt = txt[perm]
        prompt = None
        if self.prompt is not None:                                 # one prompt row per SAMPLE (a QA question differs per clip): beam-major copies
            prompt = self.prompt.repeat(beam, 1) if beam > 1 else self.prompt
        T = txt.shape[1]
        x = m._bert_embed(m._dev(txt), T, None)
        if prompt is not None:
            x = torch.cat((x, m._bert_embed(m._dev(prompt), prompt.shape[1], "prompt")), dim=1)
        Ttot = x.shape[1]
        amask = m._dev(m._bert_mask(txt, prompt, True))
        kv_range = None
        if self.kv is not None:
            kv_range = m._dev(torch.tensor([self.range] * rows, dtype=torch.int32))
        hidden = m.bert_encoder(x, amask, self.kv, kv_range, b if self.kv is not None else 0)
        idx = m._dev(torch.arange(rows, dtype=torch.int64) * Ttot + (T - 1))
        h = m.cls_transform(ops.gather_rows(hidden.reshape(-1, hidden.shape[-1]), idx))
        P = m.P
        logits = K.gemm(h, P["multimodal_encoder.embeddings.word_embeddings.weight"], bias=P["cls.decoder.bias"], out_dtype=torch.float32)
        if beam > 1:                                                # back to (sample, beam)
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(rows)
            logits = logits[m._dev(inv)]
        return logits


def log_softmax_rows(logits):
    """F.log_softmax(logits, dim=1) of fp32 rows (pretrain.py:1078): the row log-sum-exp comes from the cross-entropy kernel."""
    n, V = logits.shape
    Vpad = (V + 31) // 32 * 32
    buf = torch.zeros((n, Vpad), dtype=torch.float32, device=logits.device)
    buf[:, :V] = logits
    lse = torch.empty(n, dtype=torch.float32, device=logits.device)
    rows = torch.empty(n, dtype=torch.float32, device=logits.device)
    labels = torch.zeros(n, dtype=torch.int64, device=logits.device)
    lib.call("valor_xent_fwd", _st(), lib.DT_F32, buf.data_ptr(), labels.data_ptr(), rows.data_ptr(), lse.data_ptr(), n, V, Vpad)
    return logits - lse[:, None]


def decode_greedy(step, b, max_len):
    """VALOR.decode_greedy, model/pretrain.py:988-1028, mode 'greedy' (its logprobs are zeros in that mode)."""
    dev = step.m.device
    sents = torch.full((b, max_len), EOS, dtype=torch.long, device=dev)
    logprobs = torch.zeros((b, max_len), device=dev)
    unfinished = torch.ones(b, dtype=torch.bool, device=dev)
    state = None
    for t in range(max_len):
        wt = step.logits(state, b).max(1)[1].view(-1).long()
        unfinished = unfinished & (wt != EOS)
        wt = torch.where(unfinished, wt, torch.full_like(wt, EOS))
        sents[:, t] = wt
        w_host = wt.cpu().unsqueeze(1)                              # the next step's token ids are built on the host
        state = w_host if state is None else torch.cat((state, w_host), dim=1)
        if not bool(unfinished.any()):
            break
    return sents, logprobs


def decode_beam(step, b, beam, max_len):
    """VALOR.decode_beam / select / _adjust_tensor, model/pretrain.py:1054-1180. A beam that produced [SEP] keeps its score for every
    candidate word (:1091-1094), so its V candidates tie and the reference's sort picks among them arbitrarily: sequences are only
    defined up to that choice once a beam has ended (the tokens behind [SEP] are dropped by decode_sequence :146-164 anyway)."""
    dev = step.m.device
    seq_logprob = torch.zeros((b, 1, 1), device=dev)
    seq_mask = torch.ones((b, beam, 1), device=dev)
    outputs, selected_words, state = [], None, None
    for t in range(max_len):
        cur = 1 if t == 0 else beam
        word_logprob = log_softmax_rows(step.logits(state, b * cur)).view(b, cur, -1)
        cand = seq_logprob + word_logprob
        if t > 0:
            mask = (selected_words.view(b, cur) != EOS).float().unsqueeze(-1)
            seq_mask = seq_mask * mask
            cand = seq_mask * cand + seq_logprob.expand_as(cand) * (1 - seq_mask)
        V = cand.shape[-1]
        sel_logprob, sel_idx = torch.topk(cand.view(b, -1), beam, dim=-1, largest=True, sorted=True)     # select :1156-1159
        sel_beam = sel_idx // V
        selected_words = sel_idx - sel_beam * V
        seq_logprob = sel_logprob.unsqueeze(-1)
        seq_mask = torch.gather(seq_mask, 1, sel_beam.unsqueeze(-1))
        outputs = [torch.gather(o, 1, sel_beam.unsqueeze(-1)) for o in outputs]
        outputs.append(selected_words.unsqueeze(-1))
        w_host, beam_host = selected_words.reshape(-1, 1).cpu(), sel_beam.cpu()
        if state is not None:
            state = torch.gather(state.view(b, beam, -1), 1, beam_host.unsqueeze(-1).expand(b, beam, state.shape[1])).reshape(b * beam, -1)
            state = torch.cat((state, w_host), dim=1)
        else:
            state = w_host
    seq_logprob, sort_idx = torch.sort(seq_logprob, 1, descending=True)
    outputs = torch.gather(torch.cat(outputs, -1), 1, sort_idx.expand(b, beam, max_len))
    return outputs.contiguous()[:, 0]


def encode_for_generation(model, batch, groups):
    """The encoder half of generate_cap (pretrain.py:916-936): -> (batch size, per-layer K|V of the video/audio tokens, group key ranges)"""
    model.stage.begin_step()
    alltasks = "".join(groups)
    video_output = model.forward_video_encoder(batch["video_pixels"]) if "v" in alltasks else None
    audio_output = model.forward_audio_encoder(batch["audio_spectrograms"]) if "a" in alltasks else None
    b = (video_output if video_output is not None else audio_output).shape[0]
    kv_layers, ranges = model.cross_inputs(video_output, audio_output)
    return b, kv_layers, ranges


def stepper(model, group, b, kv_layers, ranges, prompt="caption"):
    """the per-step logits function of one query group ('tv' | 'tva' | 'ta'): stepper(...).logits(tokens so far or None, rows).
    prompt: 'caption' (the task prompt when the model uses one) or a host [b, L] tensor of prompt rows (QA: the questions)."""
    if isinstance(prompt, str):
        prompt = model.get_task_prompt(PROMPTS[prompt], b) if model.use_task_prompt else None
    return _Stepper(model, group, kv_layers, ranges, prompt, b)


@torch.no_grad()
def generate_cap(model, batch, groups, beam_size=None, max_generation_len=None):
    """VALOR.generate_cap, model/pretrain.py:914-985 -> {'generated_sequences_t_v' | '_t_va' | '_t_a' (+ 'logprobs_*' when greedy)}."""
    beam = model.beam_size if beam_size is None else beam_size
    max_len = model.max_generation_len if max_generation_len is None else max_generation_len
    was_training = model.training
    model.eval()
    try:
        b, kv_layers, ranges = encode_for_generation(model, batch, groups)
        out = {}
        for g, key in (("tv", "t_v"), ("tva", "t_va"), ("ta", "t_a")):
            if g not in groups:
                continue
            step = stepper(model, g, b, kv_layers, ranges)
            if beam > 1:
                out["generated_sequences_" + key] = decode_beam(step, b, beam, max_len)
            else:
                out["generated_sequences_" + key], out["logprobs_" + key] = decode_greedy(step, b, max_len)
        return out
    finally:
        model.train(was_training)


@torch.no_grad()
def generate_qa(model, batch, groups, prompt_cpu, beam_size=None, max_generation_len=None):
    """VALOR.generate_qa, model/pretrain.py:1366-1459: the caption decoders with the question rows as the prompt ->
    {'generated_answers_t_v' | '_t_va' | '_t_a'}. sample_num[i] consecutive question rows belong to clip i (:1378-1390): the reference
    expands the [video | audio] token rows per question; here the per-layer K|V projections are computed once per CLIP and their rows
    gathered per question (valor_gather_rows), so the projection GEMMs do not grow with the question count."""
    sample_num = [int(n) for n in batch.get("sample_num", [])]
    # the reference switches to beam search when beam_size_qa > 1 but decode_beam always searches with self.beam_size, task 'qa'
    # included (pretrain.py:1061): beam_size_qa is the switch, beam_size the width
    beam = (model.beam_size if model.beam_size_qa > 1 else 1) if beam_size is None else beam_size
    max_len = model.max_generation_len if max_generation_len is None else max_generation_len
    was_training = model.training
    model.eval()
    try:
        b, kv_layers, ranges = encode_for_generation(model, batch, groups)
        if any(n != 1 for n in sample_num):
            if len(sample_num) != b or sum(sample_num) != prompt_cpu.shape[0]:
                raise ValueError(f"sample_num {sample_num} does not describe {b} clips / {prompt_cpu.shape[0]} questions")
            from . import ops
            idx = model._dev(torch.tensor([i for i, n in enumerate(sample_num) for _ in range(n)], dtype=torch.long))
            kv_layers = [ops.gather_rows(kv.reshape(b, -1), idx).view(idx.numel(), *kv.shape[1:]) for kv in kv_layers]
            b = int(idx.numel())
        out = {}
        for g, key in (("tv", "t_v"), ("tva", "t_va"), ("ta", "t_a")):
            if g not in groups:
                continue
            step = stepper(model, g, b, kv_layers, ranges, prompt_cpu)
            out["generated_answers_" + key] = decode_beam(step, b, beam, max_len) if beam > 1 else decode_greedy(step, b, max_len)[0]
        return out
    finally:
        model.train(was_training)
