"""Tensor-level wrappers over the C-ABI (no autograd here; see ops.py).

torch is used for device memory, streams and shapes only; every computation below is a
HIP kernel in libvalor_hip.so. All tensors must live on the GPU: a CPU tensor is an error.
"""
import os

import torch

from . import lib
from .lib import ACT_NONE, DT_BF16, DT_F32

_WS = {}
_WS_BYTES = 256 << 20
# device address of the 64-bit dropout counter every dropout launch adds to its by-value offset when it RUNS (include/valor_hip.h,
# `rng_base`); 0 = by-value windows only. Set by ops.DropoutState.enable_device_base().
RNG_BASE = 0


def dt_of(t):
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype == torch.float32:
        return DT_F32
    raise TypeError(f"valor_amd kernels support bf16 / fp32 only, got {t.dtype}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _check_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise lib.ValorHipError("valor_amd kernels need GPU tensors (no CPU fallback)")


def workspace(device, nbytes=_WS_BYTES):
    """Persistent fp32 scratch (split-K partials, column-sum partials)."""
    # one per (device, stream): two compute streams (valor_amd/streams.py) must not share split-K / column-sum partials
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


def release_stream(stream):
    """forget the scratch that was allocated for `stream` (a graph-capture stream that is going away): its split-K workspace and its
    grouped-reduction queue"""
    sid = stream.cuda_stream
    for d in (_WS, ReduceQueue._queues):
        for key in [k for k in d if k[1] == sid]:
            del d[key]


# ---------------------------------------------------------------------------------------------- grouped split-K reductions
class ReduceQueue:
    """Split-K products whose reduction is deferred (valor_gemm_deferred) wait here, each with its own piece of a per-stream group
    workspace, until `flush` sums them in ONE launch per 8 (valor_gemm_reduce_group) and runs their `done` callbacks (the gradient-write
    reports the data-parallel reducer waits for: a parameter is reported only once its final write is enqueued). One queue per stream.
    Flushed when it holds 8 products, when the workspace is full, and at the end of the backward pass that filled it."""
    # products per launch: VALOR_GROUP_REDUCE = 0 (off: one reduction behind every GEMM) | 1 .. 8 (default 8, the kernel's table size)
    GROUP = max(1, min(8, int(os.environ.get("VALOR_GROUP_REDUCE", "8") or 8)))
    BYTES = 1 << 30
    PIECE = _WS_BYTES            # every product is offered what valor_gemm gets: the same slice counts, bit-identical sums
    enabled = os.environ.get("VALOR_GROUP_REDUCE", "8") != "0"
    _queues = {}
    _armed = False               # an end-of-backward flush is scheduled

    def __init__(self, device, stream):
        import ctypes
        self.stream = stream
        self.ws = torch.empty(self.BYTES, dtype=torch.uint8, device=device)
        self.blobs = (ctypes.c_char * (256 * self.GROUP))()
        self.n, self.off, self.done, self.dtype = 0, 0, [], None
        self.targets = set()       # output / row-sum buffers of the pending products (see gemm(): no two of them may be the same)

    @classmethod
    def current(cls, device):
        st = torch.cuda.current_stream(device)
        key = (device.index if device.index is not None else torch.cuda.current_device(), st.cuda_stream)
        q = cls._queues.get(key)
        if q is None:
            q = cls._queues[key] = ReduceQueue(device, st)
        return q

    def flush(self):
        if self.n:
            import ctypes
            lib.call("valor_gemm_reduce_group", self.stream.cuda_stream, self.dtype, ctypes.cast(self.blobs, ctypes.c_void_p), self.n)
        done, self.done = self.done, []
        self.n, self.off = 0, 0
        self.targets.clear()
        for fn in done:
            fn()

    @classmethod
    def flush_all(cls):
        """every queue on its own stream; the CURRENT stream then waits for the others (their sums land in the gradient arena the
        caller is about to read)"""
        cls._armed = False
        cur = torch.cuda.current_stream()
        for q in cls._queues.values():
            if q.n or q.done:
                q.flush()
                if q.stream != cur:
                    cur.wait_stream(q.stream)

    @classmethod
    def discard_stale(cls):
        """top of a forward pass: whatever is still queued (or armed) belongs to a backward pass that never finished -- an exception
        between a deferred product and the end-of-backward flush. Its partial tiles must not be summed into the NEXT step's gradients,
        and a stale `_armed` would keep the next backward from scheduling its flush."""
        stale = cls._armed or any(q.n or q.done for q in cls._queues.values())
        if stale:
            for q in cls._queues.values():
                q.n, q.off, q.done = 0, 0, []
                q.targets.clear()
            cls._armed = False
        return stale

    @classmethod
    def _arm(cls):
        if cls._armed:
            return
        try:
            from torch.autograd.variable import Variable
            Variable._execution_engine.queue_callback(cls.flush_all)
            cls._armed = True
        except RuntimeError:          # not inside a backward pass (a direct call from a test): nothing will flush later -- do it now
            cls.flush_all()


def _rowmajor(t):
    assert t.dim() == 2 and t.stride(1) == 1, "need a 2-D tensor with unit inner stride"
    return t.stride(0)


def gemm(a, b, *, trans_a=False, trans_b=False, bias=None, act=ACT_NONE, want_preact=False,
         dact_aux=None, alpha=1.0, out=None, accumulate=False, out_dtype=None, splitk=True, rowsum_out=None, rowsum_accumulate=False,
         defer_done=None, policy=None):
    """C[M,N] = epi(alpha * op(A) . op(B)^T).  A: [M,K] ([K,M] if trans_a); B: [N,K] ([K,N] if trans_b).
    defer_done (a callable): the caller does not need C before the end of the backward pass (a weight gradient accumulated into the
    arena). If the product is split along K its reduction joins the stream's ReduceQueue and `defer_done` runs when the grouped
    reduction has been enqueued; otherwise it runs at once."""
    _check_gpu(a, b, bias, dact_aux, out)
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    N, Kb = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    assert K == Kb, f"contraction mismatch {K} vs {Kb}"
    assert a.dtype == b.dtype
    lda, ldb = _rowmajor(a), _rowmajor(b)
    odt = out_dtype or a.dtype
    if out is None:
        out = torch.empty((M, N), dtype=odt, device=a.device)
    assert out.shape == (M, N) and out.dtype == odt
    ldc = _rowmajor(out)
    out_f32 = 1 if (odt == torch.float32 and a.dtype != torch.float32) else 0
    preact = None
    if want_preact:
        preact = torch.empty((M, N), dtype=odt, device=a.device)
        assert _rowmajor(preact) == ldc
    ldaux = _rowmajor(dact_aux) if dact_aux is not None else 0
    if defer_done is not None and splitk and ReduceQueue.enabled and not want_preact and policy is None:
        import ctypes
        q = ReduceQueue.current(a.device)
        dt = dt_of(a)
        # one grouped reduction adds all its products to their outputs from different workgroups of ONE launch: two pending products
        # with the same C (or the same row-sum buffer) would race on a non-atomic read-modify-write (a weight used by two Linears of a
        # shallow stack: the shared-BERT text encoder / decoder, the tied word embedding). Such a product closes the group first.
        tg = {t.data_ptr() for t in (out, rowsum_out) if t is not None}
        if q.n == q.GROUP or q.off + q.PIECE > q.BYTES or (q.n and q.dtype != dt) or (q.targets & tg):
            q.flush()
        blob = ctypes.addressof(q.blobs) + 256 * q.n
        lib.call("valor_gemm_deferred", _stream(), dt, int(trans_a), int(trans_b), M, N, K,
                 _ptr(a), lda, _ptr(b), ldb, _ptr(out), ldc, _ptr(bias), act, 0, _ptr(dact_aux), ldaux,
                 float(alpha), int(accumulate), out_f32, q.ws.data_ptr() + q.off, q.PIECE, _ptr(rowsum_out), int(rowsum_accumulate), blob)
        used = ctypes.c_int64(0)
        lib.call("valor_gemm_pending_bytes", blob, ctypes.cast(ctypes.byref(used), ctypes.c_void_p))
        if used.value:
            q.dtype = dt
            q.n += 1
            q.off += used.value
            q.done.append(defer_done)
            q.targets |= tg
            ReduceQueue._arm()
        else:
            defer_done()
        return out
    ws = workspace(a.device) if splitk else None
    if policy is not None:          # a lib.GemmPolicy: this call only, no process-global knob is touched
        import ctypes
        lib.call("valor_gemm_tuned", ctypes.addressof(policy), _stream(), dt_of(a), int(trans_a), int(trans_b), M, N, K,
                 _ptr(a), lda, _ptr(b), ldb, _ptr(out), ldc, _ptr(bias), act, _ptr(preact), _ptr(dact_aux), ldaux,
                 float(alpha), int(accumulate), out_f32, _ptr(ws), (ws.numel() * 4 if ws is not None else 0),
                 _ptr(rowsum_out), int(rowsum_accumulate))
        if defer_done is not None:
            defer_done()
        return (out, preact) if want_preact else out
    lib.call("valor_gemm", _stream(), dt_of(a), int(trans_a), int(trans_b), M, N, K,
             _ptr(a), lda, _ptr(b), ldb, _ptr(out), ldc, _ptr(bias), act, _ptr(preact), _ptr(dact_aux), ldaux,
             float(alpha), int(accumulate), out_f32, _ptr(ws), (ws.numel() * 4 if ws is not None else 0),
             _ptr(rowsum_out), int(rowsum_accumulate))
    if defer_done is not None:
        defer_done()
    return (out, preact) if want_preact else out


_INFER_POLICY = None


def infer_policy():
    """the valor_gemm_policy of inference-only products (ops.linear / ops.mlp under torch.no_grad, the decoding step of valor_amd/decode.py):
    few-row products on the weight-streaming kernel (GEMM family 5, policy key 11; VALOR_GEMM_SKINNY=0 turns it off). The training step does
    not use it: its kernels are the ones its parity evidence was collected on."""
    global _INFER_POLICY
    if _INFER_POLICY is None:
        import os
        _INFER_POLICY = lib.GemmPolicy.make(skinny=int(os.environ.get("VALOR_GEMM_SKINNY", "384")))
    return _INFER_POLICY


def gemm_fuses_rowsum(a, b, trans_a, trans_b):
    """True if valor_gemm can produce the row sums of op(A) (bias gradient) beside this GEMM."""
    if not (trans_a and a.dtype == torch.bfloat16):
        return False
    M, Kd = a.shape[1], a.shape[0]
    N = b.shape[1] if trans_b else b.shape[0]
    return lib.load().valor_gemm_kernel_for(DT_BF16, 1, int(trans_b), M, N, Kd, 0) in (3, 4)


def part_blocks():
    return lib.load().valor_ln_part_blocks()


def bdrln_fwd(x, bias, residual, gamma, beta, eps, *, p_drop=0.0, seed=0, offset=0, write_z=True, inplace_z=False,
              want_y=True, row_scale=None, rows_per_scale=0):
    """z = dropout(x + bias)/(1-p) [* row_scale[row // rows_per_scale]] + residual ; y = LN(z).  Returns (z, y, mean, rstd)."""
    _check_gpu(x, bias, residual, gamma, beta)
    cols = x.shape[-1]
    rows = x.numel() // cols
    assert x.is_contiguous() and (residual is None or residual.is_contiguous())
    z = None
    if write_z:
        z = x if inplace_z else torch.empty_like(x)
    y = torch.empty_like(x) if want_y else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if want_y else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if want_y else None
    lib.call("valor_bdrln_fwd", _stream(), dt_of(x), _ptr(x), _ptr(bias), _ptr(residual), _ptr(gamma), _ptr(beta),
             _ptr(z), _ptr(y), _ptr(mean), _ptr(rstd), rows, cols, float(eps), float(p_drop), int(seed), int(offset),
             _ptr(row_scale), int(rows_per_scale), RNG_BASE)
    return z, y, mean, rstd


def bdrln_bwd(dy, dz_in, z, mean, rstd, gamma, *, p_drop=0.0, seed=0, offset=0, want_dgamma=True, want_dbeta=True,
              want_dbias=False, separate_dx=False, sinks=(None, None, None), row_scale=None, rows_per_scale=0):
    """Returns (dx, dres, dgamma, dbeta, dbias) ; dx is dres when there is no dropout / row scale.
    sinks = (dgamma, dbeta, dbias) tensors to ACCUMULATE the parameter gradients into (the result slot is None then)."""
    ref = dy if dy is not None else dz_in
    _check_gpu(dy, dz_in, z, gamma)
    cols = ref.shape[-1]
    rows = ref.numel() // cols
    nb = part_blocks()
    ws = workspace(ref.device)
    need = 3 * nb * cols
    assert ws.numel() >= need
    pg = ws[0:nb * cols] if (want_dgamma and dy is not None) else None
    pb = ws[nb * cols:2 * nb * cols] if (want_dbeta and dy is not None) else None
    px = ws[2 * nb * cols:3 * nb * cols] if want_dbias else None
    dres = torch.empty_like(ref)
    dx = torch.empty_like(ref) if (p_drop > 0.0 or separate_dx or row_scale is not None) else dres
    lib.call("valor_bdrln_bwd", _stream(), dt_of(ref), _ptr(dy), _ptr(dz_in), _ptr(z), _ptr(mean), _ptr(rstd),
             _ptr(gamma), _ptr(dx), _ptr(dres), _ptr(pg), _ptr(pb), _ptr(px), rows, cols, float(p_drop), int(seed),
             int(offset), _ptr(row_scale), int(rows_per_scale), RNG_BASE)
    outs, args = [], []
    for part, sink in zip((pg, pb, px), sinks):
        if part is None:
            outs.append(None); args += [0, 0, 0]
        elif sink is not None:
            outs.append(None); args += [_ptr(part), _ptr(sink), 1]
        else:
            o = torch.empty(cols, dtype=ref.dtype, device=ref.device)
            outs.append(o); args += [_ptr(part), _ptr(o), 0]
    if any(a for a in args[0::3]):
        lib.call("valor_colsum_finalize3", _stream(), dt_of(ref), *args, nb, cols)
    return dx, dres, outs[0], outs[1], outs[2]


def colsum(x, out=None, accumulate=False):
    """out[cols] (+)= sum over rows of the 2-D tensor x (unit inner stride)."""
    _check_gpu(x, out)
    rows, cols = x.shape
    ld = _rowmajor(x)
    if out is None:
        out = torch.empty(cols, dtype=x.dtype, device=x.device)
        accumulate = False
    ws = workspace(x.device)
    assert ws.numel() >= part_blocks() * cols
    lib.call("valor_colsum", _stream(), dt_of(x), _ptr(x), rows, cols, ld, _ptr(ws), _ptr(out), 0, int(accumulate))
    return out


def _bsr(t):
    """(batch stride, row stride) of a [B, S, E] view with unit inner stride."""
    assert t.dim() == 3 and t.stride(2) == 1
    return t.stride(0), t.stride(1)


def attn_fwd(q, k, v, n_heads, *, mask=None, kv_range=None, kv_bmod=0, scale=None, p_drop=0.0, seed=0, offset=0, o=None, lse=None):
    """q: [B,Sq,H*64] view, k/v: [Bkv,Skv,H*64] views (unit inner stride). Returns (o [B,Sq,H*64], lse [B,H,Sq])."""
    _check_gpu(q, k, v, mask, kv_range)
    B, Sq, E = q.shape
    Skv = k.shape[1]
    assert E == n_heads * 64, "head_dim is fixed at 64"
    if scale is None:
        scale = 0.125
    if o is None:
        o = torch.empty((B, Sq, E), dtype=q.dtype, device=q.device)
    if lse is None:
        lse = torch.empty((B, n_heads, Sq), dtype=torch.float32, device=q.device)
    assert o.shape == (B, Sq, E) and lse.shape == (B, n_heads, Sq) and lse.is_contiguous()
    qb, qr = _bsr(q); kb, kr = _bsr(k); vb, vr = _bsr(v); ob, orr = _bsr(o)
    mb = mr = 0
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.dim() == 3 and mask.stride(2) == 1
        mb, mr = (mask.stride(0) if mask.shape[0] > 1 else 0), mask.stride(1)
    if kv_range is not None:
        assert kv_range.dtype == torch.int32 and kv_range.shape == (B, 2) and kv_range.is_contiguous()
    lib.call("valor_attn_fwd", _stream(), dt_of(q), _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), B, n_heads, Sq, Skv,
             qb, qr, kb, kr, vb, vr, ob, orr, _ptr(mask), mb, mr, _ptr(kv_range), int(kv_bmod), float(scale),
             float(p_drop), int(seed), int(offset), RNG_BASE)
    return o, lse


def attn_decode(q, k, v, n_heads, *, mask=None, key_row=None, scale=0.125, o=None):
    """One decoding step against K|V slots (valor_attn_decode_fwd): q [B, Sq <= 4, H*64], k / v [B, Skv <= 256, H*64] views, additive fp32
    mask [B | 1, Sq, Skv], key_row int32 [B, Skv] (slot j of sequence b is read from batch row key_row[b, j]; None: b). Returns o."""
    _check_gpu(q, k, v, mask, key_row)
    B, Sq, E = q.shape
    Skv = k.shape[1]
    assert E == n_heads * 64, "head_dim is fixed at 64"
    if o is None:
        o = torch.empty((B, Sq, E), dtype=q.dtype, device=q.device)
    qb, qr = _bsr(q); kb, kr = _bsr(k); vb, vr = _bsr(v); ob, orr = _bsr(o)
    mb = mr = 0
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.dim() == 3 and mask.stride(2) == 1
        mb, mr = (mask.stride(0) if mask.shape[0] > 1 else 0), mask.stride(1)
    rb = 0
    if key_row is not None:
        assert key_row.dtype == torch.int32 and key_row.shape == (B, Skv) and key_row.stride(1) == 1
        rb = key_row.stride(0)
    lib.call("valor_attn_decode_fwd", _stream(), dt_of(q), _ptr(q), _ptr(k), _ptr(v), _ptr(o), None, B, n_heads, Sq, Skv,
             qb, qr, kb, kr, vb, vr, ob, orr, _ptr(mask), mb, mr, _ptr(key_row), rb, float(scale))
    return o


def attn_bwd(q, k, v, o, lse, dout, n_heads, *, dq=None, dk=None, dv=None, mask=None, kv_range=None, kv_bmod=0,
             scale=None, p_drop=0.0, seed=0, offset=0, accumulate_kv=False):
    """Returns (dq, dk, dv) shaped like q, k, v (or writes into the given [B,S,H*64] views)."""
    _check_gpu(q, k, v, o, dout, mask, kv_range)
    B, Sq, E = q.shape
    Skv = k.shape[1]
    if scale is None:
        scale = 0.125
    if dq is None:
        dq = torch.empty((B, Sq, E), dtype=q.dtype, device=q.device)
    if dk is None:
        dk = torch.empty((k.shape[0], Skv, E), dtype=q.dtype, device=q.device)
    if dv is None:
        dv = torch.empty((v.shape[0], Skv, E), dtype=q.dtype, device=q.device)
    delta = torch.empty((B, n_heads, Sq), dtype=torch.float32, device=q.device)
    qb, qr = _bsr(q); kb, kr = _bsr(k); vb, vr = _bsr(v); ob, orr = _bsr(o); gb, gr = _bsr(dout)
    dqb, dqr = _bsr(dq); dkb, dkr = _bsr(dk); dvb, dvr = _bsr(dv)
    mb = mr = 0
    if mask is not None:
        mb, mr = (mask.stride(0) if mask.shape[0] > 1 else 0), mask.stride(1)
    lib.call("valor_attn_bwd", _stream(), dt_of(q), _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), _ptr(dout), _ptr(dq),
             _ptr(dk), _ptr(dv), _ptr(delta), B, n_heads, Sq, Skv, qb, qr, kb, kr, vb, vr, ob, orr, gb, gr,
             dqb, dqr, dkb, dkr, dvb, dvr, _ptr(mask), mb, mr, _ptr(kv_range), int(kv_bmod), float(scale),
             float(p_drop), int(seed), int(offset), int(accumulate_kv), RNG_BASE)
    return dq, dk, dv


def _xattn_segs(segs, kv_bmod, bwd):
    arr = (lib.XattnSeg * len(segs))()
    for i, sg in enumerate(segs):
        q, o = sg["q"], sg["o"]
        _check_gpu(q, o, sg["lse"], sg.get("kv_range"))
        a = arr[i]
        a.q, a.o, a.lse, a.kv_range = _ptr(q), _ptr(o), _ptr(sg["lse"]), _ptr(sg.get("kv_range"))
        (a.q_bs, a.q_rs), (a.o_bs, a.o_rs) = _bsr(q), _bsr(o)
        if bwd:
            do, dq = sg["dout"], sg["dq"]
            _check_gpu(do, dq)
            a.dout, a.dq = _ptr(do), _ptr(dq)
            (a.do_bs, a.do_rs), (a.dq_bs, a.dq_rs) = _bsr(do), _bsr(dq)
        a.B, a.Sq, a.seed, a.offset = q.shape[0], q.shape[1], int(sg.get("seed", 0)), int(sg.get("offset", 0))
    return arr


def _xattn_domain(segs, k, kv_bmod):
    if k.dtype != torch.bfloat16 or not (1 <= len(segs) <= 2) or k.shape[1] < 64 or kv_bmod <= 0:
        return False
    nsub = sum((sg["q"].shape[0] // kv_bmod) * ((sg["q"].shape[1] + 15) // 16) for sg in segs)
    return not (nsub > 10 or any(sg["q"].shape[0] % kv_bmod for sg in segs) or os.environ.get("VALOR_ATTN_XFUSED", "1") == "0")


def _xattn_call(name, *args):
    """the fused cross-attention entries validate their own domain (strides that are multiples of 8 elements, 16-byte bases, < 2 GiB
    per operand, <= ten 16-row query sub-tiles, the VALOR_ATTN_XFUSED switch as the C side parses it) and answer VALOR_ERR_ARG (-1)
    BEFORE anything is launched when a call falls outside it: that is "not covered" (False -> the caller runs the per-pass kernels),
    not a failure. Every other non-zero code is one."""
    rc = getattr(lib.load(), name)(*args)
    if rc == -1:
        return False
    if rc != 0:
        raise lib.ValorHipError(f"{name} failed with code {rc}")
    return True


def cross_attn_fwd_fused(segs, k, v, n_heads, kv_bmod, *, scale=0.125, p_drop=0.0):
    """Forward of up to two decoder passes that attend to the same K|V in ONE launch (csrc/attention_xu.hip): K, V read once.
    segs: list of dict(q, o [B, T, E] views, lse [B, H, T], kv_range int32 [B, 2] or None, seed, offset); o and lse are written.
    Returns False (nothing launched) outside the fused kernel's domain -- the caller runs attn_fwd per pass."""
    if not _xattn_domain(segs, k, kv_bmod):
        return False
    import ctypes
    arr = _xattn_segs(segs, kv_bmod, False)
    kb, kr = _bsr(k); vb, vr = _bsr(v)
    return _xattn_call("valor_cross_attn_fwd_fused", _stream(), DT_BF16, ctypes.cast(arr, ctypes.c_void_p), len(segs), _ptr(k), _ptr(v),
                       n_heads, k.shape[1], int(kv_bmod), kb, kr, vb, vr, float(scale), float(p_drop), RNG_BASE)


def cross_attn_bwd_fused(segs, k, v, dk, dv, n_heads, kv_bmod, *, scale=0.125, p_drop=0.0):
    """Backward of up to two decoder passes that attend to the same K|V in ONE launch (csrc/attention_xu.hip): dK|dV are written once.
    segs: list of dict(q, o, lse, dout, dq [B, T, E] views / lse [B, H, T]; kv_range int32 [B, 2] or None; seed, offset).
    Returns False (nothing launched) when the shape is outside the fused kernel's domain -- the caller runs attn_bwd per pass."""
    if not _xattn_domain(segs, k, kv_bmod):
        return False
    arr = _xattn_segs(segs, kv_bmod, True)
    kb, kr = _bsr(k); vb, vr = _bsr(v); dkb, dkr = _bsr(dk); dvb, dvr = _bsr(dv)
    import ctypes
    return _xattn_call("valor_cross_attn_bwd_fused", _stream(), DT_BF16, ctypes.cast(arr, ctypes.c_void_p), len(segs), _ptr(k), _ptr(v), _ptr(dk),
                       _ptr(dv), n_heads, k.shape[1], int(kv_bmod), kb, kr, vb, vr, dkb, dkr, dvb, dvr, float(scale), float(p_drop), RNG_BASE)


# ---------------------------------------------------------------------------------------------- VideoSwin
def win_attn_fwd(qkv, geo, table, n_heads, B):
    """3-D shifted-window attention (head_dim 32) in place on the fused QKV rows [B*rows_per_sample, 3C].
    geo: dict(rowmap int32 [nW*N], rel int32 [N], rel_inv int32 [relc+1], label uint8 [nW*N] | None, nW, N, relc, rows).
    Returns (o, lse)."""
    _check_gpu(qkv, table, geo["rowmap"], geo["rel"], geo["label"])
    C = n_heads * 32
    assert qkv.is_contiguous() and qkv.shape == (B * geo["rows"], 3 * C) and table.is_contiguous() and table.shape[1] == n_heads
    o = torch.empty((qkv.shape[0], C), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B * geo["nW"], n_heads, geo["N"]), dtype=torch.float32, device=qkv.device)
    lib.call("valor_win_attn_fwd", _stream(), dt_of(qkv), _ptr(qkv), _ptr(o), _ptr(lse), _ptr(geo["rowmap"]), _ptr(geo["rel"]),
             _ptr(geo["label"]), _ptr(table), B, geo["nW"], geo["N"], n_heads, table.shape[0], geo["relc"], geo["rows"], 32 ** -0.5)
    return o, lse


def win_attn_bwd(qkv, o, lse, dout, geo, table, n_heads, B, dtable=None):
    """Returns (dqkv, dtable); dtable given -> accumulated into (and None returned in its place)."""
    _check_gpu(qkv, o, dout, table, dtable)
    assert dout.is_contiguous() and dout.shape == o.shape
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    acc = dtable is not None
    if dtable is None:
        dtable = torch.empty_like(table)
    need = lib.load().valor_win_attn_workspace_floats(B, geo["nW"], geo["N"], n_heads)
    if need < 0:
        raise lib.ValorHipError("window attention workspace too large")
    ws = workspace(qkv.device, max(_WS_BYTES, need * 4))
    lib.call("valor_win_attn_bwd", _stream(), dt_of(qkv), _ptr(qkv), _ptr(o), _ptr(lse), _ptr(dout), _ptr(dqkv), _ptr(delta),
             _ptr(geo["rowmap"]), _ptr(geo["rel"]), _ptr(geo["rel_inv"]), _ptr(geo["label"]), _ptr(table), _ptr(dtable), int(acc), _ptr(ws),
             ws.numel() * 4,
             B, geo["nW"], geo["N"], n_heads, table.shape[0], geo["relc"], geo["rows"], 32 ** -0.5)
    return dqkv, (None if acc else dtable)


def patchify3d(video_f32, P, out_dtype):
    """video [B, F, C, H, W] fp32 -> [B*F*(H/P)*(W/P), C*2*P*P] rows of PatchEmbed3D's conv (one zero frame appended)."""
    _check_gpu(video_f32)
    assert video_f32.dtype == torch.float32 and video_f32.is_contiguous()
    B, F, C, H, W = video_f32.shape
    out = torch.empty((B * F * (H // P) * (W // P), C * 2 * P * P), dtype=out_dtype, device=video_f32.device)
    lib.call("valor_patchify3d", _stream(), dt_of(out), _ptr(video_f32), _ptr(out), B, F, C, H, W, P)
    return out


def group_mean_fwd(x2d, X):
    _check_gpu(x2d)
    assert x2d.is_contiguous() and x2d.shape[0] % X == 0
    out = torch.empty((x2d.shape[0] // X, x2d.shape[1]), dtype=x2d.dtype, device=x2d.device)
    lib.call("valor_group_mean_fwd", _stream(), dt_of(x2d), _ptr(x2d), _ptr(out), out.shape[0], X, x2d.shape[1])
    return out


def group_mean_bwd(dout, X):
    _check_gpu(dout)
    dout = dout.contiguous()
    din = torch.empty((dout.shape[0] * X, dout.shape[1]), dtype=dout.dtype, device=dout.device)
    lib.call("valor_group_mean_bwd", _stream(), dt_of(dout), _ptr(dout), _ptr(din), dout.shape[0], X, dout.shape[1])
    return din
