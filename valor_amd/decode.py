"""Caption generation on the native decoder: VALOR.generate_cap / decode_greedy / decode_beam / get_logits
(model/pretrain.py:914-1189), caption_type 'unimlm' | 'lm'.

The reference feeds [CLS] + generated tokens + [MASK] through the whole decoder again at every step (its cache path is disabled when the
decoder has cross-attention, pretrain.py:890-896, bert.py:848-850, and broken behind that, bert.py:807). Two things are cached here:

  * the video/audio K|V projections of the 12 layers are computed ONCE per clip (VALOR.cross_inputs) and shared by every step, every
    group and every beam;
  * the text rows' self-attention K|V (`DecodeSession`, SURVEY 8 row f4 "a working KV cache"). The generation mask is causal over the
    text and every text row sees the prompt rows, which see only themselves (bert.py:879-885): a position's hidden states never change
    once its token is fixed. A decoding step therefore runs TWO rows per sequence -- the token chosen by the previous step at
    position t and the [MASK] at t + 1 ('lm': one row) -- against per-layer K|V slots [prompt | text positions]; the prompt rows run
    once per clip. The step has static shapes and no host-built tensor (the step counter, the key-validity mask and the slot indices
    live on the device), so it is captured ONCE per geometry as a hipGraph and replayed: ~250 launches per step leave the host.
    Sessions (static buffers + graph) are kept per model and geometry and reused across batches and query groups.
    VALOR_KV_CACHE=0 selects the re-run-everything path (`_Stepper`, also the teacher-forced logits the parity tests use);
    VALOR_DECODE_GRAPH=0 runs the cached step eagerly.

Beam rows are kept in the reference's (sample, beam) order on the host side; the decoder sees them beam-major (row = beam * b + sample)
so that a row's cross-attention K|V is row % b -- the kv_bmod addressing the training passes use -- instead of beam copies of K|V
(pretrain.py:1133-1139 expands video_input / audio_input beam_size times).

Everything here is inference: torch.no_grad, dropout off regardless of model.training (the reference calls it under model.eval())."""
import os

import torch

from . import kernels as K, lib, ops
from .model.valor import PROMPTS, _BlockKV

BOS, EOS, MASK = 101, 102, 103          # [CLS] / [SEP] / [MASK] of bert-base-uncased, model/modeling.py:669-671


def _st():
    return K._stream()


class _Stepper:
    """get_logits + forward_cap_single(compute_loss=False), pretrain.py:1031-1051,882-900, for one query group."""

    def __init__(self, model, group, kv_layers, ranges, prompt_cpu, b):
        self.m, self.kv, self.b, self.prompt = model, kv_layers, b, prompt_cpu
        self.range = list(ranges[group]) if kv_layers is not None else None
        self.blocks = isinstance(kv_layers, _BlockKV)

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
        if self.blocks:                                             # a cross-attention block per modality: the group is a tuple of modalities
            kv_range = tuple(self.range)
        elif self.kv is not None:
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


NEG = -10000.0                           # the additive mask value of BertModel.forward (bert.py:885)


def kv_cache_enabled():
    return os.environ.get("VALOR_KV_CACHE", "1") != "0"


class DecodeSession:
    """Static buffers + the captured decoding step of one geometry: R = b * beam sequences, J new rows per step (2: token + [MASK],
    'unimlm'; 1: 'lm'), P prompt slots, `max_len` generated tokens, the per-layer [video | audio] K|V shape.

    Slots of a layer's self-attention K|V, per sequence: [0, P) the prompt rows (written once per clip by `begin_group`), P + i the
    text position i. Step t writes the token's K|V to slot P + t and the [MASK]'s to P + t + 1 (overwritten by the next step's token);
    `kmask` [R, L] holds 0 / -10000 per written slot (a key whose token id is 0 is masked, bert.py:857,885: prompt padding, and a
    generated id 0), the step's attention mask opens slots <= P + t for the token row and P + t + 1 besides for the [MASK] row.
    Rows are beam-major (row = beam * b + sample); a beam step first moves every sequence's slots to the row that continues it
    (`parent`)."""

    def __init__(self, model, b, beam, P, max_len, kv_layers):
        m = self.m = model
        dev, dt, E = m.device, m.dtype, m.spec.hidden
        self.b, self.beam, self.R, self.P, self.max_len = b, beam, b * beam, P, max_len
        self.J = 1 if m.caption_type == "lm" else 2
        self.L = P + max_len + self.J - 1
        self.layers = m.spec.layers
        R, L, J = self.R, self.L, self.J
        i64 = dict(dtype=torch.int64, device=dev)
        # beam search: every sequence continues the row `parent` of the previous step -- two slot buffers, the step gathers the rows of the
        # one into the other with ONE launch of the row-gather kernel over all layers (a torch index_select + copy back was 0.42 of a 3.8 ms
        # step). VALOR_BEAM_TABLE=1: the slots never move (a token slot is written once and read-only afterwards), a [R, L] table of row
        # numbers follows the beams and the attention kernel reads slot j of sequence r from row slot_row[r, j]: half the slot memory, no
        # gather traffic -- and measured EQUAL (profiles/r06_generation_beam_table{,_off}.json: 553.9 vs 556.6 captions/s), so not the default.
        self.table = beam > 1 and os.environ.get("VALOR_BEAM_TABLE", "0") != "0"
        self.caches = [torch.zeros((self.layers, R, L, 2 * E), dtype=dt, device=dev) for _ in range(2 if beam > 1 and not self.table else 1)]
        self.rows32 = torch.arange(R, device=dev, dtype=torch.int32)
        self.slot_row = self.rows32[:, None].expand(R, L).contiguous() if self.table else None
        self.cache = self.caches[0]                    # the buffer that holds the sequences' slots now
        self.step_i = 0                                # steps of the current group so far (host side: which buffer is live)
        self.layer_base = (torch.arange(self.layers, device=dev, dtype=torch.int64) * R)[:, None]
        self.kmask = torch.full((R, L), NEG, dtype=torch.float32, device=dev)
        self.amask = torch.empty((R, J, L), dtype=torch.float32, device=dev)
        self.t = torch.zeros(1, **i64)
        self.tok = torch.full((R,), BOS, **i64)
        self.parent = torch.arange(R, **i64)
        self.ids = torch.full((R, J), MASK, **i64)
        self.slot = torch.arange(L, **i64)
        self.jidx = torch.arange(J, **i64)
        self.kv = None
        if kv_layers is not None:
            self.kv = [torch.empty_like(kv) for kv in kv_layers]
            self.cross_range = torch.zeros((R, 2), dtype=torch.int32, device=dev)
        self.graphs, self.logits_of, self.pool = {}, {}, None
        self.logits_pad = self.slots_buf = None
        self.fused_head = dev.type == "cuda" and os.environ.get("VALOR_DECODE_PROLOGUE", "1") != "0"
        self.eager_steps = 0
        self.use_graph = dev.type == "cuda" and os.environ.get("VALOR_DECODE_GRAPH", "1") != "0"

    @staticmethod
    def key(model, b, beam, P, max_len, kv_layers):
        kvs = None if kv_layers is None else (tuple(kv_layers[0].shape), len(kv_layers))
        return (b, beam, P, max_len, kvs, model.caption_type)

    def begin_batch(self, kv_layers):
        """the clips' per-layer [video | audio] K|V into the session's static buffers (once per batch: every group reads them)"""
        if kv_layers is not None:
            for dst, src in zip(self.kv, kv_layers):
                dst.copy_(src)

    def begin_group(self, key_range, prompt_cpu):
        """reset the text slots; run the prompt rows (one set per clip, bert.py:879-885: they attend to themselves only) and keep their
        per-layer K|V in slots [0, P) of every beam's row"""
        m, b, P, E = self.m, self.b, self.P, self.m.spec.hidden
        self.step_i, self.cache = 0, self.caches[0]
        self.t.zero_()
        self.tok.fill_(BOS)
        self.parent.copy_(self.layer_base.new_tensor(range(self.R)))
        if self.table:
            self.slot_row.copy_(self.rows32[:, None].expand(self.R, self.L))
        self.kmask.fill_(NEG)
        if self.kv is not None:
            self.cross_range.copy_(m._dev(torch.tensor([list(key_range)] * self.R, dtype=torch.int32)))
        if P:
            valid = m._dev((prompt_cpu != 0).to(torch.float32))                                 # [b, P]
            pm = (1.0 - valid) * NEG
            self.kmask.view(self.beam, b, self.L)[:, :, :P] = pm[None]
            pmask = pm[:, None, :].expand(b, P, P).contiguous()
            xp = m._bert_embed(m._dev(prompt_cpu), P, "prompt")
            cache = self.cache.view(self.layers, self.beam, b, self.L, 2 * E)

            def prefill(i, qkv):
                cache[i, :, :, :P] = qkv[:, :, E:][None]
                return ops.self_attention(qkv, m.spec.heads, pmask, 0.0)
            m.bert_encoder(xp, pmask, self.kv, self.cross_range[:b] if self.kv is not None else None, b if self.kv is not None else 0,
                           self_attn=prefill)

    # ------------------------------------------------------------------ one decoding step (eager, and what the graph captures)
    def _self_attn(self, i, qkv):
        E, c = self.m.spec.hidden, self.cache[i]
        c.index_copy_(1, self.slots_new, qkv[:, :, E:])
        if self.table:
            return K.attn_decode(qkv[:, :, :E], c[:, :, :E], c[:, :, E:], self.m.spec.heads, mask=self.amask, key_row=self.slot_row, scale=0.125)
        o, _ = K.attn_fwd(qkv[:, :, :E], c[:, :, :E], c[:, :, E:], self.m.spec.heads, mask=self.amask, scale=0.125)
        return o

    def _body(self, cur=0):
        """cur: which of the two slot buffers holds the sequences' state when the step starts (beam search; greedy has one)"""
        m, P_, J = self.m, self.m.P, self.J
        e = "multimodal_encoder.embeddings."
        self.cache = self.caches[cur]
        if self.table:                                              # every sequence continues the row `parent` of the previous step
            self.slot_row.copy_(self.slot_row.index_select(0, self.parent))
            self.kmask.copy_(self.kmask.index_select(0, self.parent))
        elif self.beam > 1:
            src, dst = self.caches[cur], self.caches[1 - cur]
            idx = (self.layer_base + self.parent[None, :]).reshape(-1)
            words = src[0, 0].numel() * src.element_size() // 4      # a sequence's slots of one layer as 32-bit words (16-byte accesses)
            lib.call("valor_gather_rows", _st(), lib.DT_F32, src.data_ptr(), idx.data_ptr(), dst.data_ptr(), idx.numel(), words, words)
            self.kmask.copy_(self.kmask.index_select(0, self.parent))
            self.cache = dst
        Wd, Pe, Ty = P_[e + "word_embeddings.weight"], P_[e + "position_embeddings.weight"], P_[e + "token_type_embeddings.weight"]
        if self.fused_head and Wd.dtype == Pe.dtype == Ty.dtype == m.dtype and Wd.is_contiguous() and Pe.is_contiguous() and Ty.is_contiguous():
            # embeddings before their LayerNorm, the step's mask rows, its slots: one launch (valor_decode_prologue) for the ~22 below
            x = torch.empty((self.R, J, Wd.shape[1]), dtype=m.dtype, device=Wd.device)
            if self.slots_buf is None:
                self.slots_buf = torch.zeros(J, dtype=torch.int64, device=Wd.device)
            lib.call("valor_decode_prologue", _st(), K.dt_of(x), self.tok.data_ptr(), self.t.data_ptr(), Wd.data_ptr(), Pe.data_ptr(), Ty.data_ptr(),
                     MASK, self.R, J, Wd.shape[1], self.P, self.L, NEG, self.kmask.data_ptr(), self.amask.data_ptr(), x.data_ptr(),
                     self.slots_buf.data_ptr())
            self.slots_new = self.slots_buf
            if self.table:
                self.slot_row.index_copy_(1, self.slots_new, self.rows32[:, None].expand(self.R, J).contiguous())
            x = ops.layer_norm(x, P_[e + "LayerNorm.weight"], P_[e + "LayerNorm.bias"], 1e-12)
            return self._tail(x)
        self.ids[:, 0] = self.tok
        pos = self.t + self.jidx
        # BertEmbeddings (bert.py:190-218) at positions t, t + 1: the embedding kernel's arithmetic (fp32 sum, one rounding)
        x = (P_[e + "word_embeddings.weight"][self.ids].float() + P_[e + "position_embeddings.weight"][pos].float()[None]
             + P_[e + "token_type_embeddings.weight"][0].float()).to(m.dtype)
        x = ops.layer_norm(x, P_[e + "LayerNorm.weight"], P_[e + "LayerNorm.bias"], 1e-12)
        slot_t = self.t + self.P
        self.kmask.index_copy_(1, slot_t, torch.where(self.tok != 0, 0.0, NEG)[:, None])
        row_a = torch.where(self.slot <= slot_t, self.kmask, NEG)
        self.amask[:, 0] = row_a
        if J == 2:
            self.amask[:, 1] = torch.where(self.slot == slot_t + 1, 0.0, row_a)
        self.slots_new = slot_t + self.jidx
        if self.table:                                              # this step's slots are the row's own
            self.slot_row.index_copy_(1, self.slots_new, self.rows32[:, None].expand(self.R, J).contiguous())
        return self._tail(x)

    def _tail(self, x):
        m, P_, J = self.m, self.m.P, self.J
        e = "multimodal_encoder.embeddings."
        hidden = m.bert_encoder(x, None, self.kv, self.cross_range if self.kv is not None else None, self.b if self.kv is not None else 0,
                                self_attn=self._self_attn)
        h = m.cls_transform(hidden[:, J - 1].contiguous())
        if self.logits_pad is None:                                   # rows of a zero-padded static buffer (first, eager, step of the session):
            V = P_[e + "word_embeddings.weight"].shape[0]             # log_softmax_rows reads them in place
            self.logits_pad = torch.zeros((h.shape[0], (V + 31) // 32 * 32), dtype=torch.float32, device=h.device)[:, :V]
        logits = self.logits_pad
        K.gemm(h, P_[e + "word_embeddings.weight"], bias=P_["cls.decoder.bias"], out=logits, out_dtype=torch.float32, policy=K.infer_policy())
        self.t += 1
        return logits

    def _capture(self, cur):
        from . import graphs
        dev = self.m.device
        ctx = graphs._capture_ctx(dev)
        with torch.cuda.stream(ctx["stream"]):         # the capture stream's kernel scratch must not come out of the graph's pool
            K.workspace(dev)
            K.ReduceQueue.current(dev)
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        g = torch.cuda.CUDAGraph()
        mode = "thread_local" if (torch.distributed.is_available() and torch.distributed.is_initialized()) else "global"
        torch.cuda.current_stream(dev).synchronize()
        with torch.cuda.graph(g, pool=self.pool, stream=ctx["stream"], capture_error_mode=mode):
            self.logits_of[cur] = self._body(cur)
        self.graphs[cur] = g

    def step(self, tok=None, parent=None):
        """tok int64 [R] (device; None: [CLS], the first step), parent int64 [R] (beam search) -> fp32 logits [R, vocab] of the step's
        last row (a static buffer when the step is a graph replay: consume it before the next step)"""
        if tok is not None:
            self.tok.copy_(tok)
        if parent is not None:
            self.parent.copy_(parent)
        cur = self.step_i % len(self.caches)
        self.step_i += 1
        if cur not in self.graphs:
            if not self.use_graph or self.eager_steps < 1:
                self.eager_steps += 1
                return self._body(cur)
            self._capture(cur)
        self.graphs[cur].replay()
        self.cache = self.caches[(1 - cur) % len(self.caches)]
        return self.logits_of[cur]


MAX_SESSIONS = 4


def session(model, b, beam, P, max_len, kv_layers):
    """the model's DecodeSession of this geometry (most recently used last; at most MAX_SESSIONS are kept)"""
    pool = model.__dict__.setdefault("_decode_sessions", {})
    key = DecodeSession.key(model, b, beam, P, max_len, kv_layers)
    s = pool.pop(key, None)
    if s is None:
        while len(pool) >= MAX_SESSIONS:
            pool.pop(next(iter(pool)))
        s = DecodeSession(model, b, beam, P, max_len, kv_layers)
    pool[key] = s
    return s


def release_sessions(model):
    """drop the decoding sessions (their K|V slots, the copies of the clips' K|V and the captured graphs)"""
    model.__dict__.pop("_decode_sessions", None)


def row_lse(logits):
    """log-sum-exp of fp32 rows (the cross-entropy kernel's) and the zero-padded buffer the rows sit in"""
    n, V = logits.shape
    Vpad = (V + 31) // 32 * 32
    if logits.stride() == (Vpad, 1) and logits.data_ptr() % 16 == 0:       # the decoding session's rows already sit in a zero-padded buffer
        buf = logits
    else:
        buf = torch.zeros((n, Vpad), dtype=torch.float32, device=logits.device)
        buf[:, :V] = logits
    lse = torch.empty(n, dtype=torch.float32, device=logits.device)
    rows = torch.empty(n, dtype=torch.float32, device=logits.device)
    labels = torch.zeros(n, dtype=torch.int64, device=logits.device)
    lib.call("valor_xent_fwd", _st(), lib.DT_F32, buf.data_ptr(), labels.data_ptr(), rows.data_ptr(), lse.data_ptr(), n, V, Vpad)
    return lse, buf


def log_softmax_rows(logits):
    """F.log_softmax(logits, dim=1) of fp32 rows (pretrain.py:1078): the row log-sum-exp comes from the cross-entropy kernel."""
    return logits - row_lse(logits)[0][:, None]


def beam_select_enabled():
    return os.environ.get("VALOR_BEAM_SELECT", "1") != "0"


def beam_select(logits, b, cur, beam, seq_logprob, seq_mask, beam_major, lse="kernel", lse_out=None):
    """candidate scores + `select` (pretrain.py:1080-1098,1156-1159) in one launch (valor_beam_select): logits fp32 [b * cur, V], rows
    beam-major (row = k * b + s) or sample-major; seq_logprob [b, cur | 1, 1], seq_mask [b, cur, 1] (1: open) -> (values, indices) [b, beam].
    lse: 'kernel' (the rows' log-sum-exp inside the launch; left in lse_out [b * cur] if given) or 'xent' (the cross-entropy kernel's)"""
    V = logits.shape[1]
    dev = logits.device
    if lse == "xent":
        lse_t, buf = row_lse(logits)
    else:
        lse_t, buf = None, (logits if logits.stride(1) == 1 else logits.contiguous())
    sl = seq_logprob.reshape(b, -1).expand(b, cur).contiguous()
    sm = seq_mask.reshape(b, -1)[:, :cur].contiguous()
    val = torch.empty((b, beam), dtype=torch.float32, device=dev)
    idx = torch.empty((b, beam), dtype=torch.int64, device=dev)
    rs_s, rs_k = (1, b) if beam_major else (cur, 1)
    lib.call("valor_beam_select", _st(), buf.data_ptr(), buf.stride(0), rs_s, rs_k, None if lse_t is None else lse_t.data_ptr(), sl.data_ptr(),
             sm.data_ptr(), b, cur, V, beam, val.data_ptr(), idx.data_ptr(), None if lse_out is None else lse_out.data_ptr())
    return val, idx


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


def decode_greedy_cached(sess, b, max_len):
    """decode_greedy on a DecodeSession: the tokens stay on the device (the next step's input is this step's argmax); whether every row
    has ended is asked every eighth step (a finished row keeps producing [SEP], pretrain.py:1005-1010: the result is the same)."""
    dev = sess.m.device
    sents = torch.full((b, max_len), EOS, dtype=torch.long, device=dev)
    logprobs = torch.zeros((b, max_len), device=dev)
    unfinished = torch.ones(b, dtype=torch.bool, device=dev)
    tok = None
    for t in range(max_len):
        wt = sess.step(tok).max(1)[1]
        unfinished = unfinished & (wt != EOS)
        tok = torch.where(unfinished, wt, EOS)
        sents[:, t] = tok
        if t % 8 == 7 and t + 1 < max_len and not bool(unfinished.any()):
            break
    return sents, logprobs


def decode_beam_cached(sess, b, beam, max_len):
    """decode_beam on a DecodeSession (rows beam-major): the selection arithmetic of pretrain.py:1054-1180 on the device, the chosen
    beams as the next step's `parent` rows. Step 0 runs `beam` identical copies of every sequence and reads the first."""
    dev = sess.m.device
    seq_logprob = torch.zeros((b, 1, 1), device=dev)
    seq_mask = torch.ones((b, beam, 1), device=dev)
    base = torch.arange(b, device=dev)
    outputs = torch.zeros((b, beam, max_len), dtype=torch.int64, device=dev)
    selected_words, tok, parent = None, None, None
    fused = beam_select_enabled() and beam <= 8
    for t in range(max_len):
        cur = 1 if t == 0 else beam
        logits = sess.step(tok, parent)
        V = logits.shape[-1]
        if t > 0:
            mask = (selected_words.view(b, cur) != EOS).float().unsqueeze(-1)
            seq_mask = seq_mask * mask
        if fused:
            # (lse='kernel' -- the log-sum-exp inside the launch -- measured equal: 64 workgroups make three more passes over their rows)
            sel_logprob, sel_idx = beam_select(logits[:b * cur], b, cur, beam, seq_logprob, seq_mask, beam_major=True, lse="xent")
        else:
            word_logprob = log_softmax_rows(logits[:b * cur]).view(cur, b, -1).transpose(0, 1)
            cand = seq_logprob + word_logprob
            if t > 0:
                cand = seq_mask * cand + seq_logprob.expand_as(cand) * (1 - seq_mask)
            sel_logprob, sel_idx = torch.topk(cand.reshape(b, -1), beam, dim=-1, largest=True, sorted=True)   # select :1156-1159
        sel_beam = sel_idx // V
        selected_words = sel_idx - sel_beam * V
        seq_logprob = sel_logprob.unsqueeze(-1)
        seq_mask = torch.gather(seq_mask, 1, sel_beam.unsqueeze(-1))
        if t > 0:                                                    # the words so far follow their beams (:1101; one gather, not one per step)
            outputs = torch.gather(outputs, 1, sel_beam.unsqueeze(-1).expand(b, beam, max_len))
        outputs[:, :, t] = selected_words
        parent = (sel_beam.t() * b + base[None]).reshape(-1)        # row (k, s) continues row (sel_beam[s, k], s)
        tok = selected_words.t().reshape(-1)
    seq_logprob, sort_idx = torch.sort(seq_logprob, 1, descending=True)
    outputs = torch.gather(outputs, 1, sort_idx.expand(b, beam, max_len))
    return outputs.contiguous()[:, 0]


def _decode_groups(model, groups, b, kv_layers, ranges, prompt, beam, max_len):
    """{group: (sequences, logprobs | None)} for the query groups present, through the K|V-cached session or the re-run path"""
    if isinstance(prompt, str):
        prompt = model.get_task_prompt(PROMPTS[prompt], b) if model.use_task_prompt else None
    out = {}
    sess = None
    if kv_cache_enabled() and not isinstance(kv_layers, _BlockKV):      # (a block per modality, bert.py:459-496: the re-run path)
        sess = session(model, b, beam, 0 if prompt is None else prompt.shape[1], max_len, kv_layers)
        sess.begin_batch(kv_layers)
    for g in ("tv", "tva", "ta"):
        if g not in groups:
            continue
        if sess is not None:
            sess.begin_group(ranges[g] if kv_layers is not None else None, prompt)
            out[g] = (decode_beam_cached(sess, b, beam, max_len), None) if beam > 1 else decode_greedy_cached(sess, b, max_len)
        else:
            step = _Stepper(model, g, kv_layers, ranges, prompt, b)
            out[g] = (decode_beam(step, b, beam, max_len), None) if beam > 1 else decode_greedy(step, b, max_len)
    return out


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
        for g, (seq, lp) in _decode_groups(model, groups, b, kv_layers, ranges, "caption", beam, max_len).items():
            key = {"tv": "t_v", "tva": "t_va", "ta": "t_a"}[g]
            out["generated_sequences_" + key] = seq
            if lp is not None:
                out["logprobs_" + key] = lp
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
            rows_of = lambda kvs: None if kvs is None else [ops.gather_rows(kv.reshape(b, -1), idx).view(idx.numel(), *kv.shape[1:]) for kv in kvs]
            kv_layers = _BlockKV(rows_of(kv_layers.v), rows_of(kv_layers.a)) if isinstance(kv_layers, _BlockKV) else rows_of(kv_layers)
            b = int(idx.numel())
        res = _decode_groups(model, groups, b, kv_layers, ranges, prompt_cpu, beam, max_len)
        return {"generated_answers_" + {"tv": "t_v", "tva": "t_va", "ta": "t_a"}[g]: seq for g, (seq, _lp) in res.items()}
    finally:
        model.train(was_training)
