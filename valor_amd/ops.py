"""Autograd operators of the VALOR step: every forward AND backward below is a HIP kernel call
through the C-ABI (valor_amd.kernels / valor_amd.lib). torch.autograd only sequences them.

There is no eager / CPU implementation of any op here: tensors must be on the GPU and
libvalor_hip.so must be built, otherwise lib.ValorHipError is raised.
"""
import math
import os
import weakref

import torch
from torch.autograd import Function

from . import kernels as K
from . import lib
from .lib import ACT_NONE

_st = K._stream
_p = K._ptr
_dt = K.dt_of


class DropoutState:
    """Counter-based dropout RNG bookkeeping: every op draws a fresh (seed, offset) window; kernels
    re-generate masks in backward from the same window (nothing is stored).

    Two modes. Host mode (default): `offset` grows for the life of the process and travels BY VALUE into every launch. Device mode
    (enable_device_base): the by-value offsets restart at 0 at every step (begin_step) -- they are the same numbers step after step,
    which is what a captured graph needs -- and a 64-bit counter in device memory, advanced by STEP_STRIDE at every begin_step, is
    added by the kernels when they RUN (include/valor_hip.h `rng_base`). Windows of different steps never overlap as long as a step
    draws fewer than STEP_STRIDE counters (2^40; the base configuration draws ~2^33)."""
    seed = 1234
    offset = 0
    base = None                    # int64 [1] device tensor (device mode) or None
    begun = False                  # begin_step() ran and no forward pass has consumed it yet (VALOR._forward_groups checks)
    STEP_STRIDE = 1 << 40

    @classmethod
    def reset(cls, seed, offset=0):
        cls.seed, cls.offset = int(seed), int(offset)
        if cls.base is not None:
            cls.base.zero_()

    @classmethod
    def enable_device_base(cls, device):
        """switch to device mode (idempotent); the counter starts at 0"""
        if cls.base is None or cls.base.device != torch.device(device):
            cls.base = torch.zeros(1, dtype=torch.int64, device=device)
            K.RNG_BASE = cls.base.data_ptr()
        cls.offset = 0
        return cls.base

    _retired = []                  # counters of earlier device-mode sessions: their ADDRESS may be baked into graphs that still exist

    @classmethod
    def disable_device_base(cls):
        if cls.base is not None:
            cls._retired.append(cls.base)        # 8 bytes; never freed, so a stale captured graph reads a valid (if meaningless) word
        cls.base = None
        K.RNG_BASE = 0

    @classmethod
    def begin_step(cls):
        """top of a training step (device mode only): by-value offsets restart, the device counter moves to this step's range. One tiny
        kernel on the current stream, ahead of everything the step launches (side streams fork from this stream afterwards)."""
        if cls.base is not None:
            cls.offset = 0
            cls.base.add_(cls.STEP_STRIDE)
            cls.begun = True

    @classmethod
    def draw(cls, n_elements):
        """window for a Philox4x32 stream (4 elements per counter): LayerNorm-side dropout"""
        off = cls.offset
        cls.offset += (int(n_elements) + 3) // 4 + 1
        return cls.seed, off

    @classmethod
    def draw_elems(cls, n_elements):
        """window for the per-element attention hash (one index per probability)"""
        off = cls.offset
        cls.offset += int(n_elements) + 1
        return cls.seed, off


def _2d(x):
    return x.reshape(-1, x.shape[-1])


# VALOR_FUSE_GLUE=0: layer_norm_stream / tap_rows fall back to separate nodes whose gradients autograd adds (A/B of the glue kernels)
FUSE_GLUE = os.environ.get("VALOR_FUSE_GLUE", "1") != "0"


class GradSink:
    """Parameter gradients are accumulated by the producing kernel straight into the flat gradient arena
    (wgrad GEMM epilogue `C += ...`, column-sum finalize with accumulate) instead of being returned to autograd,
    which would materialise a temporary and launch one `grad += tmp` kernel per parameter (AccumulateGrad).
    `listener(name)` (set by valor_amd.dist.Reducer) is told about every such write so that used-parameter tracking
    and bucket launches work exactly as with autograd's post-accumulate hooks."""
    enabled = True
    listener = None
    recorder = None        # a list while a backward is being CAPTURED into a graph (valor_amd/graphs.py): names are recorded, not reported
    live_hooks = set()     # names whose post-accumulate hook ran in the CURRENT (outer) backward pass (dist.Reducer; see graphs._Replay)


class GradSlot:
    """Meeting point for the gradients of ONE activation that feeds several ops (the shared cross K/V buffer of a decoder
    layer used by the caption and mlm passes; the [video|audio] decoder input feeding 12 K/V projections). The first
    backward to run writes its gradient into a fresh buffer and returns it to autograd; later ones accumulate INTO that
    buffer inside their own kernel (GEMM epilogue `C +=`, attention dK/dV +=) and return nothing. Autograd still sees the
    right total whichever subset of the consumers takes part in a backward pass, and the separate add kernels are gone."""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


def _sink(p):
    if not GradSink.enabled or p is None or getattr(p, "_arena_name", None) is None or not p.requires_grad:
        return None
    view = getattr(p, "_sink_view", None)        # a reshaped view of a parameter (param_view): the same reshape of its gradient
    return view if view is not None else p.grad


def param_view(p, *shape):
    """A reshaped view of an arena parameter (the conv kernels used as GEMM weights: [W, 3, p, p] -> [W, 3 p p]) that still accumulates
    its gradient straight into the arena: without this the wgrad GEMM returns a temporary, autograd reshapes it back and AccumulateGrad
    adds it -- on the stream the parameter's node was created on, which for the side-stream encoders is not the producing stream."""
    v = p.view(*shape)
    if GradSink.enabled and getattr(p, "_arena_name", None) is not None and p.grad is not None and p.requires_grad:
        v._arena_name = p._arena_name
        v._sink_view = p.grad.view(*shape)
    return v


def _sunk(p):
    if GradSink.recorder is not None:
        GradSink.recorder.append(p._arena_name)
    elif GradSink.listener is not None:
        GradSink.listener(p._arena_name)


# ------------------------------------------------------------------------------------------------
class CheckpointFn(Function):
    """Activation checkpointing of one layer (the reference's `checkpointing` option: torch.utils.checkpoint around every resblock /
    encoder layer / VideoSwin block, clip.py:208-209, transformer.py:163-164, videoswin.py:234-241,448-449, bert.py:510-513): the forward
    keeps the layer's INPUTS only, the backward runs the layer again -- with ops.DropoutState rewound to the offsets the first run drew,
    so the regenerated dropout masks are the forward's -- and back-propagates through that second run. Parameter gradients take their
    usual route (GradSink kernels accumulate into the arena, the reducer is told by the recomputed layer's backward)."""
    first_pass = 0

    @staticmethod
    def forward(ctx, fn, *args):
        ctx.fn = fn
        ctx.off0 = DropoutState.offset
        CheckpointFn.first_pass += 1           # the layer must run the kernels its second (differentiated) run will: bit-identical activations
        try:
            with torch.no_grad():
                outs = fn(*args)
        finally:
            CheckpointFn.first_pass -= 1
        ctx.single = not isinstance(outs, tuple)
        ctx.save_for_backward(*args)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        args = ctx.saved_tensors
        cur = DropoutState.offset
        DropoutState.offset = ctx.off0
        ins = [a.detach().requires_grad_(ctx.needs_input_grad[i + 1]) for i, a in enumerate(args)]
        with torch.enable_grad():
            outs = ctx.fn(*ins)
        DropoutState.offset = cur
        outs = (outs,) if ctx.single else outs
        pairs = [(o, g) for o, g in zip(outs, gouts) if g is not None and o.requires_grad]
        torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
        return (None,) + tuple(i.grad for i in ins)


def checkpoint(fn, *tensors):
    """fn(*tensors) -> tensor | tuple of tensors, with only `tensors` kept for backward (see CheckpointFn)"""
    return CheckpointFn.apply(fn, *tensors)


# ------------------------------------------------------------------------------------------------
class LinearFn(Function):
    """y = act(x W^T + b).  W: [N,K] (or, with w_is_kn, a [K,N] matrix used as x @ W: CLIP projections
    clip.py:237,329 / pretrain.py:90-91). Backward: dX = dY.W (dgrad GEMM), dW = dY^T.X (wgrad GEMM,
    split-K), db = column sums."""

    @staticmethod
    def forward(ctx, x, w, b, act, w_is_kn, slot=None, out=None):
        ctx.slot = slot
        x2 = _2d(x)
        want_pre = act != ACT_NONE
        kw = {} if out is None else {"out": _2d(out)}        # a caller-owned (static) output buffer: nothing is allocated here
        if w_is_kn:
            res = K.gemm(x2, w, trans_b=True, bias=b, act=act, want_preact=want_pre, **kw)
        else:
            res = K.gemm(x2, w, bias=b, act=act, want_preact=want_pre, **kw)
        y, pre = res if want_pre else (res, None)
        ctx.save_for_backward(x2, w, pre)
        ctx.act, ctx.w_is_kn, ctx.has_b, ctx.xshape = act, w_is_kn, b is not None, x.shape
        ctx.params = (w, b)
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        x2, w, pre = ctx.saved_tensors
        dy2 = _2d(dy.contiguous())
        if ctx.act != ACT_NONE:
            du = torch.empty_like(dy2)
            lib.call("valor_dact_mul", _st(), _dt(dy2), _p(dy2), _p(pre), _p(du), dy2.numel(), ctx.act)
            dy2 = du
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            slot = ctx.slot
            if slot is not None and slot.buf is not None:          # a sibling already produced a gradient buffer: add into it
                out2 = _2d(slot.buf)
                K.gemm(dy2, w, out=out2, accumulate=True) if ctx.w_is_kn else K.gemm(dy2, w, trans_b=True, out=out2, accumulate=True)
            else:
                dx = (K.gemm(dy2, w) if ctx.w_is_kn else K.gemm(dy2, w, trans_b=True)).view(ctx.xshape)
                if slot is not None:
                    slot.buf = dx
        pw, pb = ctx.params
        sb = _sink(pb) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        fused_b = False
        if ctx.needs_input_grad[1]:
            sw = _sink(pw)
            kw = dict(out=sw, accumulate=True) if sw is not None else {}
            if ctx.w_is_kn:
                if sw is not None:
                    kw["defer_done"] = lambda: _sunk(pw)          # the split-K reduction joins the layer's group (K.ReduceQueue)
                dw = K.gemm(x2, dy2, trans_a=True, trans_b=True, **kw)
            else:
                if sb is not None and K.gemm_fuses_rowsum(dy2, x2, True, True):      # bias gradient on the matrix pipe beside dW
                    kw.update(rowsum_out=sb, rowsum_accumulate=True); fused_b = True
                if sw is not None:
                    kw["defer_done"] = (lambda: (_sunk(pw), _sunk(pb))) if fused_b else (lambda: _sunk(pw))
                dw = K.gemm(dy2, x2, trans_a=True, trans_b=True, **kw)
            if sw is not None:
                dw = None
        if ctx.has_b and ctx.needs_input_grad[2]:
            if fused_b:
                if sw is None:
                    _sunk(pb)
            elif sb is not None:
                K.colsum(dy2, out=sb, accumulate=True); _sunk(pb)
            else:
                db = K.colsum(dy2)
        return dx, dw, db, None, None, None, None


def linear(x, w, b=None, act=ACT_NONE, w_is_kn=False, grad_slot=None, out=None):
    if not torch.is_grad_enabled() and not CheckpointFn.first_pass:      # _inference(): no pre-activation copy, few-row products on family 5
        kw = {} if out is None else {"out": _2d(out)}
        y = K.gemm(_2d(x), w, trans_b=w_is_kn, bias=b, act=act, policy=K.infer_policy(), **kw)
        return y.view(*x.shape[:-1], y.shape[-1])
    return LinearFn.apply(x, w, b, act, w_is_kn, grad_slot, out)


_MLP_SAVES_DERIV = os.environ.get("VALOR_MLP_DERIV", "1") != "0"      # 0: keep the pre-activation and evaluate act' in the dgrad (A/B runs)


class MlpFn(Function):
    """y = act(x W1^T + b1) W2^T + b2  (BertIntermediate+BertOutput.dense bert.py:403-406,417; CLIP mlp
    clip.py:178-182; AST FeedForward transformer.py:141-142; fine-weight MLP pretrain.py:104-112).
    The activation derivative is fused into the dgrad GEMM epilogue: dU = (dY.W2) * act'(u). What the forward keeps for it is act'(u)
    itself (ACT_DERIV), evaluated beside act(u) from the same sigmoid / erf: the backward epilogue is one multiply."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act, slot=None):
        ctx.slot = slot
        x2 = _2d(x)
        if _MLP_SAVES_DERIV:
            act = act | lib.ACT_DERIV
        h, u = K.gemm(x2, w1, bias=b1, act=act, want_preact=True)
        y = K.gemm(h, w2, bias=b2)
        ctx.save_for_backward(x2, w1, w2, u, h)
        ctx.act, ctx.xshape = act, x.shape
        ctx.params = (w1, b1, w2, b2)
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, u, h = ctx.saved_tensors
        dy2 = _2d(dy.contiguous())
        du = K.gemm(dy2, w2, trans_b=True, act=ctx.act, dact_aux=u)
        pw1, pb1, pw2, pb2 = ctx.params

        def wgrad_bgrad(dyy, xx, pw, pb):
            """dW (+ db fused on the matrix pipe where the kernel offers it), accumulated straight into the arena"""
            sw = _sink(pw)
            sb = _sink(pb) if pb is not None else None
            kw, fused = {}, False
            if sb is not None and K.gemm_fuses_rowsum(dyy, xx, True, True):
                kw.update(rowsum_out=sb, rowsum_accumulate=True); fused = True
            if sw is None:
                dw = K.gemm(dyy, xx, trans_a=True, trans_b=True, **kw)
                if fused:
                    _sunk(pb)
            else:
                # the split-K reduction of this product joins the layer's group (K.ReduceQueue); the writes are reported when it is enqueued
                done = (lambda: (_sunk(pw), _sunk(pb))) if fused else (lambda: _sunk(pw))
                K.gemm(dyy, xx, trans_a=True, trans_b=True, out=sw, accumulate=True, defer_done=done, **kw); dw = None
            if pb is None:
                return dw, None
            if fused:
                return dw, None
            if sb is None:
                return dw, K.colsum(dyy)
            K.colsum(dyy, out=sb, accumulate=True); _sunk(pb)
            return dw, None

        # (a frozen weight -- VALOR's frozen_multimodal / frozen_vision -- needs no gradient GEMM at all)
        dw2, db2 = wgrad_bgrad(dy2, h, pw2, pb2) if (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]) else (None, None)
        dx = None
        if ctx.needs_input_grad[0]:
            slot = ctx.slot
            if slot is not None and slot.buf is not None:          # the residual path already produced x's gradient buffer: add into it
                K.gemm(du, w1, trans_b=True, out=_2d(slot.buf), accumulate=True)
            else:
                dx = K.gemm(du, w1, trans_b=True).view(ctx.xshape)
                if slot is not None:
                    slot.buf = dx
        dw1, db1 = wgrad_bgrad(du, x2, pw1, pb1) if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) else (None, None)
        return dx, dw1, db1, dw2, db2, None, None


def _inference():
    """no autograd graph is being built (torch.no_grad: evaluation, generation) and this is not the first run of a checkpointed layer, which
    must run the kernels its second, differentiated run will (bit-identical activations)"""
    return not torch.is_grad_enabled() and not CheckpointFn.first_pass


def mlp(x, w1, b1, w2, b2, act, grad_slot=None):
    if _inference():       # no second [rows, inter] output of the first GEMM (the saved act'(u): 620 MB per ViT layer at the bench shape)
        y = K.gemm(K.gemm(_2d(x), w1, bias=b1, act=act, policy=K.infer_policy()), w2, bias=b2, policy=K.infer_policy())
        return y.view(*x.shape[:-1], y.shape[-1])
    return MlpFn.apply(x, w1, b1, w2, b2, act, grad_slot)


# ------------------------------------------------------------------------------------------------
class BdrLnFn(Function):
    """(z, y) = fused bias + dropout + residual + LayerNorm (see csrc/layernorm.hip).
    mode flags: want_z -> return the pre-LN sum z (pre-LN residual stream); otherwise only y."""

    @staticmethod
    def forward(ctx, x, bias, residual, gamma, beta, eps, p_drop, want_z, row_scale=None, rows_per_scale=0, res_slot=None):
        x = x.contiguous()
        ctx.set_materialize_grads(False)
        ctx.res_slot = res_slot          # GradSlot of `residual` (post-LN blocks: the same activation also feeds the sub-layer's first GEMM)
        seed = off = 0
        if p_drop > 0:
            seed, off = DropoutState.draw(x.numel())
        plain = bias is None and residual is None and p_drop == 0 and row_scale is None
        # plain + want_z (layer_norm_stream): "z" is x itself -- the first block of a pre-LN stack, whose input feeds both the block's
        # LayerNorm and its residual add: the two gradients then meet inside this node's backward kernel (dz_in) instead of in an add
        z, y, mean, rstd = K.bdrln_fwd(x, bias, residual.contiguous() if residual is not None else None, gamma, beta,
                                       eps, p_drop=p_drop, seed=seed, offset=off, write_z=not plain, row_scale=row_scale,
                                       rows_per_scale=rows_per_scale)
        ctx.save_for_backward(x if plain else z, mean, rstd, gamma)
        ctx.rs = (row_scale, rows_per_scale)
        ctx.cfg = (p_drop, seed, off, bias is not None, residual is not None, beta is not None)
        ctx.params = (gamma, beta, bias)
        if want_z:
            return (x.view_as(x) if plain else z), y
        return y

    @staticmethod
    def backward(ctx, *grads):
        zs, mean, rstd, gamma = ctx.saved_tensors
        p_drop, seed, off, has_b, has_r, has_beta = ctx.cfg
        if len(grads) == 2:
            dz_in, dy = grads
        else:
            dz_in, dy = None, grads[0]
        dy = dy.contiguous() if dy is not None else None
        dz_in = dz_in.contiguous() if dz_in is not None else None
        if dy is None and dz_in is None:
            return (None,) * 11
        if dy is None:   # only the residual stream was used downstream
            dy_eff, mean_e, rstd_e, z_e = None, None, None, None
        else:
            dy_eff, mean_e, rstd_e, z_e = dy, mean, rstd, zs
        pg, pbeta, pbias = ctx.params
        sinks = (_sink(pg) if dy_eff is not None else None, _sink(pbeta) if dy_eff is not None else None, _sink(pbias))
        dx, dres, dg, dbeta, dbias = K.bdrln_bwd(dy_eff, dz_in, z_e, mean_e, rstd_e, gamma, p_drop=p_drop, seed=seed,
                                                 offset=off, want_dgamma=gamma is not None, want_dbeta=has_beta,
                                                 want_dbias=has_b, sinks=sinks, row_scale=ctx.rs[0], rows_per_scale=ctx.rs[1],
                                                 separate_dx=has_r and ctx.res_slot is not None)
        # separate_dx: without dropout / row scale the kernel returns ONE tensor as dx and dres. With a res_slot the residual's other
        # consumer accumulates into dres in its own kernel (C +=) while dx is still the incoming gradient of x's producer -- the same
        # memory if they alias (a one-GEMM sub-layer would read dY from the buffer it accumulates dX into). Two tensors, always.
        assert not (has_r and ctx.res_slot is not None) or dx.data_ptr() != dres.data_ptr()
        for prm, sk, want in ((pg, sinks[0], gamma is not None), (pbeta, sinks[1], has_beta), (pbias, sinks[2], has_b)):
            if sk is not None and want:
                _sunk(prm)
        if has_r and ctx.res_slot is not None and ctx.res_slot.buf is None:
            ctx.res_slot.buf = dres          # this node runs before the residual's other consumer: that one adds into dres in its own kernel
        return dx, dbias, (dres if has_r else None), dg, dbeta, None, None, None, None, None, None


def layer_norm(x, gamma, beta, eps):
    return BdrLnFn.apply(x, None, None, gamma, beta, eps, 0.0, False)


def layer_norm_stream(x, gamma, beta, eps):
    """(x, LayerNorm(x)) as ONE autograd node: for an x that is also the residual stream of the block it enters (the first block of a
    pre-LN stack, clip.py:194-214 / transformer.py:156-170). The residual gradient comes back as the first output's gradient and is
    added inside the LayerNorm backward kernel; autograd's own add of two [rows, E] tensors (three passes over them) is gone."""
    if not FUSE_GLUE:
        return x, layer_norm(x, gamma, beta, eps)
    return BdrLnFn.apply(x, None, None, gamma, beta, eps, 0.0, True)


def bias_dropout_residual_ln(x, bias, residual, gamma, beta, eps, p_drop, want_z, row_scale=None, rows_per_scale=0, res_slot=None):
    """row_scale fp32 [rows / rows_per_scale]: per-sample stochastic-depth factor on (x + bias) (videoswin.py:40-49).
    res_slot: GradSlot shared with the OTHER consumer of `residual` (a linear / mlp given the same slot as grad_slot): that consumer's
    dgrad GEMM accumulates into this node's residual gradient instead of autograd adding two tensors."""
    return BdrLnFn.apply(x, bias, residual, gamma, beta, eps, p_drop, want_z, row_scale, rows_per_scale, res_slot)


class BiasDropResFn(Function):
    """z = dropout(x + bias)/(1-p) + residual  without a LayerNorm (last residual add of a pre-LN block)."""

    @staticmethod
    def forward(ctx, x, bias, residual, p_drop, row_scale=None, rows_per_scale=0):
        x = x.contiguous()
        seed = off = 0
        if p_drop > 0:
            seed, off = DropoutState.draw(x.numel())
        z, _, _, _ = K.bdrln_fwd(x, bias, residual.contiguous() if residual is not None else None, None, None, 0.0,
                                 p_drop=p_drop, seed=seed, offset=off, write_z=True, want_y=False, row_scale=row_scale,
                                 rows_per_scale=rows_per_scale)
        ctx.rs = (row_scale, rows_per_scale)
        ctx.cfg = (p_drop, seed, off, bias is not None, residual is not None)
        ctx.params = (bias,)
        return z

    @staticmethod
    def backward(ctx, dz):
        p_drop, seed, off, has_b, has_r = ctx.cfg
        sb = _sink(ctx.params[0]) if has_b else None
        dx, dres, _, _, dbias = K.bdrln_bwd(None, dz.contiguous(), None, None, None, None, p_drop=p_drop, seed=seed,
                                            offset=off, want_dgamma=False, want_dbeta=False, want_dbias=has_b,
                                            sinks=(None, None, sb), row_scale=ctx.rs[0], rows_per_scale=ctx.rs[1])
        if sb is not None:
            _sunk(ctx.params[0])
        return dx, dbias, (dres if has_r else None), None, None, None


def bias_dropout_residual(x, bias, residual, p_drop, row_scale=None, rows_per_scale=0):
    return BiasDropResFn.apply(x, bias, residual, p_drop, row_scale, rows_per_scale)


# ------------------------------------------------------------------------------------------------
class SelfAttnFn(Function):
    """Self-attention on the fused QKV GEMM output qkv [B,S,3E] (q | k | v column blocks)."""

    @staticmethod
    def forward(ctx, qkv, n_heads, mask, p_drop):
        B, S, E3 = qkv.shape
        E = E3 // 3
        seed = off = 0
        if p_drop > 0:
            seed, off = DropoutState.draw_elems(B * n_heads * S * S)
        q, k, v = qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:]
        o, lse = K.attn_fwd(q, k, v, n_heads, mask=mask, scale=1.0 / math.sqrt(64), p_drop=p_drop, seed=seed, offset=off)
        ctx.save_for_backward(qkv, o, lse, mask)
        ctx.cfg = (n_heads, p_drop, seed, off)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse, mask = ctx.saved_tensors
        n_heads, p_drop, seed, off = ctx.cfg
        E = qkv.shape[2] // 3
        dqkv = torch.empty_like(qkv)
        q, k, v = qkv[:, :, :E], qkv[:, :, E:2 * E], qkv[:, :, 2 * E:]
        K.attn_bwd(q, k, v, o, lse, do.contiguous(), n_heads, dq=dqkv[:, :, :E], dk=dqkv[:, :, E:2 * E], dv=dqkv[:, :, 2 * E:],
                   mask=mask, scale=1.0 / math.sqrt(64), p_drop=p_drop, seed=seed, offset=off)
        return dqkv, None, None, None


def self_attention(qkv, n_heads, mask=None, p_drop=0.0):
    return SelfAttnFn.apply(qkv, n_heads, mask, p_drop)


class StaticGen:
    """Generation counter of a set of STATIC buffers that one forward writes and its backward reads (VALOR.project_cross_kv: the per-layer
    K|V tensors and their gradient buffers live outside the caching allocator because they cross streams). The owner bumps `gen` whenever
    it rewrites the buffers; an autograd node that saved one of them remembers the generation it saw and refuses to run backward on a
    newer one -- a second training forward before the first backward would otherwise give silently wrong gradients (the write is a raw
    kernel into `out`, invisible to autograd's version counters)."""
    REGISTRY = weakref.WeakValueDictionary()          # data_ptr of a registered buffer -> its StaticGen (gone with its owner: a model that is
                                                      # dropped without release_static_kv() leaves no entry behind for a recycled address)

    def __init__(self):
        self.gen = 0
        self.ptrs = []

    def register(self, tensors):
        self.release()
        self.ptrs = [t.data_ptr() for t in tensors]
        for q in self.ptrs:
            StaticGen.REGISTRY[q] = self

    def release(self):
        for q in self.ptrs:
            StaticGen.REGISTRY.pop(q, None)
        self.ptrs = []

    @staticmethod
    def seen(t):
        g = StaticGen.REGISTRY.get(t.data_ptr())
        return None if g is None else (g, g.gen)

    @staticmethod
    def check(tag):
        if tag is not None and tag[0].gen != tag[1]:
            raise RuntimeError("valor_amd: the static cross-attention K|V buffers were rewritten by a later training forward before this backward "
                               "ran (one forward per backward with VALOR_KV_STREAM=1; set VALOR_KV_STREAM=0 to keep several graphs alive)")


class CrossAttnFn(Function):
    """Modality-grouped cross-attention: q [B,T,E]; kv [Bkv,Skv,2E] = ONE projected K|V set of the
    concatenated [video | audio] tokens shared by every query group; kv_range[b] = (start, len)."""

    @staticmethod
    def forward(ctx, q, kv, n_heads, kv_range, kv_bmod, p_drop, slot=None):
        ctx.slot = slot
        B, T, E = q.shape
        seed = off = 0
        if p_drop > 0:
            seed, off = DropoutState.draw_elems(B * n_heads * T * kv.shape[1])
        k, v = kv[:, :, :E], kv[:, :, E:]
        q = q.contiguous()
        o, lse = K.attn_fwd(q, k, v, n_heads, kv_range=kv_range, kv_bmod=kv_bmod, scale=1.0 / math.sqrt(64),
                            p_drop=p_drop, seed=seed, offset=off)
        ctx.save_for_backward(q, kv, o, lse, kv_range)
        ctx.cfg = (n_heads, kv_bmod, p_drop, seed, off)
        ctx.static_gen = StaticGen.seen(kv)
        return o

    @staticmethod
    def backward(ctx, do):
        StaticGen.check(ctx.static_gen)
        q, kv, o, lse, kv_range = ctx.saved_tensors
        n_heads, kv_bmod, p_drop, seed, off = ctx.cfg
        E = q.shape[2]
        slot = ctx.slot
        acc = slot is not None and slot.buf is not None
        dkv = slot.buf if acc else torch.empty_like(kv)
        dq, _, _ = K.attn_bwd(q, kv[:, :, :E], kv[:, :, E:], o, lse, do.contiguous(), n_heads, dk=dkv[:, :, :E],
                              dv=dkv[:, :, E:], kv_range=kv_range, kv_bmod=kv_bmod, scale=1.0 / math.sqrt(64),
                              p_drop=p_drop, seed=seed, offset=off, accumulate_kv=acc)
        if slot is not None and not acc:
            slot.buf = dkv
        return dq, (None if acc else dkv), None, None, None, None, None


def cross_attention(q, kv, n_heads, kv_range=None, kv_bmod=0, p_drop=0.0, grad_slot=None):
    return CrossAttnFn.apply(q, kv, n_heads, kv_range, kv_bmod, p_drop, grad_slot)


# ------------------------------------------------------------------------------------------------
# Row-batched decoder passes: several decoder passes (caption groups with T = 32, mlm with T = 42 ...) share the same
# weights, so their rows are stacked into ONE [R, E] activation matrix for every GEMM / LayerNorm of the layer; only the
# attention calls (different sequence length / mask / K-V ranges per pass) work on row segments of it, in place.
class SegSelfAttnFn(Function):
    """qkv2d [R, 3E]; segs = [(r0, B, T, mask), ...] with rows r0 .. r0+B*T of pass-major [B, T] layout."""

    @staticmethod
    def forward(ctx, qkv2d, n_heads, segs, p_drop):
        R, E3 = qkv2d.shape
        E = E3 // 3
        o = torch.empty((R, E), dtype=qkv2d.dtype, device=qkv2d.device)
        lses, rng = [], []
        for (r0, B, T, mask) in segs:
            seed = off = 0
            if p_drop > 0:
                seed, off = DropoutState.draw_elems(B * n_heads * T * T)
            v3 = qkv2d[r0:r0 + B * T].view(B, T, E3)
            lse = torch.empty((B, n_heads, T), dtype=torch.float32, device=qkv2d.device)
            K.attn_fwd(v3[:, :, :E], v3[:, :, E:2 * E], v3[:, :, 2 * E:], n_heads, mask=mask, scale=1.0 / math.sqrt(64), p_drop=p_drop,
                       seed=seed, offset=off, o=o[r0:r0 + B * T].view(B, T, E), lse=lse)
            lses.append(lse); rng.append((seed, off))
        ctx.save_for_backward(qkv2d, o, *lses)
        ctx.cfg = (n_heads, segs, p_drop, rng)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv2d, o, *lses = ctx.saved_tensors
        n_heads, segs, p_drop, rng = ctx.cfg
        E = qkv2d.shape[1] // 3
        do = do.contiguous()
        dqkv = torch.empty_like(qkv2d)
        for (r0, B, T, mask), lse, (seed, off) in zip(segs, lses, rng):
            v3 = qkv2d[r0:r0 + B * T].view(B, T, 3 * E)
            d3 = dqkv[r0:r0 + B * T].view(B, T, 3 * E)
            K.attn_bwd(v3[:, :, :E], v3[:, :, E:2 * E], v3[:, :, 2 * E:], o[r0:r0 + B * T].view(B, T, E), lse,
                       do[r0:r0 + B * T].view(B, T, E), n_heads, dq=d3[:, :, :E], dk=d3[:, :, E:2 * E], dv=d3[:, :, 2 * E:],
                       mask=mask, scale=1.0 / math.sqrt(64), p_drop=p_drop, seed=seed, offset=off)
        return dqkv, None, None, None


def seg_self_attention(qkv2d, n_heads, segs, p_drop=0.0):
    return SegSelfAttnFn.apply(qkv2d, n_heads, segs, p_drop)


_XFUSED_FWD = os.environ.get("VALOR_ATTN_XFUSED_FWD", "0") == "1"


class SegCrossAttnFn(Function):
    """q2d [R, E]; kv [Bkv, Skv, 2E] shared; segs = [(r0, B, T, kv_range, kv_bmod), ...].  In backward the first segment
    writes dK|dV and the others accumulate into the same buffer inside the kernel."""

    @staticmethod
    def forward(ctx, q2d, kv, n_heads, segs, p_drop, dkv_buf=None):
        ctx.dkv_buf = dkv_buf
        R, E = q2d.shape
        q2d = q2d.contiguous()
        o = torch.empty((R, E), dtype=q2d.dtype, device=q2d.device)
        lses, rng = [], []
        for (r0, B, T, kv_range, kv_bmod) in segs:
            seed = off = 0
            if p_drop > 0:
                seed, off = DropoutState.draw_elems(B * n_heads * T * kv.shape[1])
            lses.append(torch.empty((B, n_heads, T), dtype=torch.float32, device=q2d.device)); rng.append((seed, off))
        fused = False
        # every pass in ONE launch (K, V read once): opt-in. Measured at the decoder's geometry (profiles/r04_attn_xu_ab_fwd.json) the two
        # per-pass launches take 228 us and the fused one 300 us: the forward has no dK|dV read-modify-write to save, the per-pass kernels
        # are not HBM bound, and the D-split needs three workgroup barriers per 64-key tile.
        if _XFUSED_FWD and len(segs) <= 2 and len({sg[4] for sg in segs}) == 1:
            fs = [dict(q=q2d[r0:r0 + B * T].view(B, T, E), o=o[r0:r0 + B * T].view(B, T, E), lse=lse, kv_range=kv_range, seed=seed, offset=off)
                  for (r0, B, T, kv_range, kv_bmod), lse, (seed, off) in zip(segs, lses, rng)]
            fused = K.cross_attn_fwd_fused(fs, kv[:, :, :E], kv[:, :, E:], n_heads, segs[0][4] or segs[0][1], scale=1.0 / math.sqrt(64),
                                           p_drop=p_drop)
        for (r0, B, T, kv_range, kv_bmod), lse, (seed, off) in zip(segs, lses, rng):
            if fused:
                break
            K.attn_fwd(q2d[r0:r0 + B * T].view(B, T, E), kv[:, :, :E], kv[:, :, E:], n_heads, kv_range=kv_range, kv_bmod=kv_bmod,
                       scale=1.0 / math.sqrt(64), p_drop=p_drop, seed=seed, offset=off, o=o[r0:r0 + B * T].view(B, T, E), lse=lse)
        ctx.save_for_backward(q2d, kv, o, *lses)
        ctx.cfg = (n_heads, segs, p_drop, rng)
        ctx.static_gen = StaticGen.seen(kv)
        return o

    @staticmethod
    def backward(ctx, do):
        StaticGen.check(ctx.static_gen)
        q2d, kv, o, *lses = ctx.saved_tensors
        n_heads, segs, p_drop, rng = ctx.cfg
        E = q2d.shape[1]
        do = do.contiguous()
        dq = torch.empty_like(q2d)
        # a static per-layer buffer when the K|V projection lives on the side stream (a tensor that crosses streams through the caching
        # allocator cannot be reused until the other stream has passed it: the pool grows and hipMalloc stalls the step)
        dkv = ctx.dkv_buf if ctx.dkv_buf is not None else torch.empty_like(kv)
        # every pass in ONE launch when the fused kernel covers the shape (K, V read once, dK|dV written once)
        if len(segs) <= 2 and len({sg[4] for sg in segs}) == 1:
            fs = [dict(q=q2d[r0:r0 + B * T].view(B, T, E), o=o[r0:r0 + B * T].view(B, T, E), dout=do[r0:r0 + B * T].view(B, T, E),
                       dq=dq[r0:r0 + B * T].view(B, T, E), lse=lse, kv_range=kv_range, seed=seed, offset=off)
                  for (r0, B, T, kv_range, kv_bmod), lse, (seed, off) in zip(segs, lses, rng)]
            if K.cross_attn_bwd_fused(fs, kv[:, :, :E], kv[:, :, E:], dkv[:, :, :E], dkv[:, :, E:], n_heads, segs[0][4] or segs[0][1],
                                      scale=1.0 / math.sqrt(64), p_drop=p_drop):
                return dq, dkv, None, None, None, None
        for i, ((r0, B, T, kv_range, kv_bmod), lse, (seed, off)) in enumerate(zip(segs, lses, rng)):
            sl = slice(r0, r0 + B * T)
            K.attn_bwd(q2d[sl].view(B, T, E), kv[:, :, :E], kv[:, :, E:], o[sl].view(B, T, E), lse, do[sl].view(B, T, E), n_heads,
                       dq=dq[sl].view(B, T, E), dk=dkv[:, :, :E], dv=dkv[:, :, E:], kv_range=kv_range, kv_bmod=kv_bmod,
                       scale=1.0 / math.sqrt(64), p_drop=p_drop, seed=seed, offset=off, accumulate_kv=i > 0)
        return dq, dkv, None, None, None, None


def seg_cross_attention(q2d, kv, n_heads, segs, p_drop=0.0, dkv_buf=None):
    return SegCrossAttnFn.apply(q2d, kv, n_heads, segs, p_drop, dkv_buf)


_ROWS_HWM = {}


def _rows_with_slack(n, cols, dtype, device):
    """[n, cols] rows of a buffer allocated for the LARGEST row count seen so far plus ~12 % (rounded to 512 rows): the number of masked
    rows -- hence the size of the [rows, vocabulary] logits, 100+ MB -- changes from step to step. Sized exactly, every new maximum is a
    device allocation in the middle of a step and a smaller request splits whatever large cached block is free; with a monotone
    request size the caching allocator hands the same block back every step. (The device allocations bench.py still counts inside its
    timed region are not these: `timed_region.new_segments_mb` shows two [117 376, 768] and one [100 864, 3072] bf16 tensors -- blocks
    held by their cross-stream use records while the host runs two steps ahead of the GPU right after the warm-up sync.)"""
    key = (cols, dtype, str(device))
    rows = max(_ROWS_HWM.get(key, 0), (n + n // 8 + 511) // 512 * 512)
    _ROWS_HWM[key] = rows
    return torch.empty((rows, cols), dtype=dtype, device=device)[:n]


class DecoderXentSegFn(Function):
    """per-segment mean CE over ONE tied-decoder GEMM: rows of segment i are h[sum(n[:i]) : sum(n[:i+1])]; returns one
    loss per segment (caption / mlm passes share the prediction head, modeling.py:245-254, pretrain.py:444,498)."""

    @staticmethod
    def forward(ctx, h, w_emb, dec_bias, labels, seg_rows, smoothing=0.0):
        n, V = h.shape[0], w_emb.shape[0]
        Vpad = (V + 31) // 32 * 32
        buf = _rows_with_slack(n, Vpad, h.dtype, h.device)
        K.gemm(h, w_emb, bias=dec_bias, out=buf[:, :V])
        loss_rows = torch.empty(n, dtype=torch.float32, device=h.device)
        lse = torch.empty(n, dtype=torch.float32, device=h.device)
        ctx.smoothing = float(smoothing)          # LabelSmoothing (pretrain.py:46-61): the caption finetune loss with config.label_smoothing > 0
        lib.call("valor_xent_smooth_fwd", _st(), _dt(h), _p(buf), _p(labels), _p(loss_rows), _p(lse), n, V, Vpad, ctx.smoothing)
        losses, r0 = [], 0
        for nr in seg_rows:
            l = torch.empty((), dtype=torch.float32, device=h.device)
            lib.call("valor_mean_f32", _st(), _p(loss_rows[r0:r0 + nr]), nr, _p(l))
            losses.append(l); r0 += nr
        ctx.save_for_backward(h, w_emb, labels, lse, buf)
        ctx.V, ctx.seg_rows = V, tuple(seg_rows)
        ctx.params = (w_emb, dec_bias)
        return tuple(losses)

    @staticmethod
    def backward(ctx, *dlosses):
        h, w_emb, labels, lse, buf = ctx.saved_tensors
        V, Vpad = ctx.V, buf.shape[1]
        r0 = 0
        for nr, dl in zip(ctx.seg_rows, dlosses):
            g = dl.to(torch.float32).contiguous() if dl is not None else torch.zeros((), dtype=torch.float32, device=h.device)
            lib.call("valor_xent_smooth_bwd", _st(), _dt(h), _p(buf[r0:r0 + nr]), _p(labels[r0:r0 + nr]), _p(lse[r0:r0 + nr]), _p(g), 1.0 / nr,
                     nr, V, Vpad, ctx.smoothing)
            r0 += nr
        dlog = buf[:, :V]
        dh = K.gemm(dlog, w_emb, trans_b=True)
        pw, pb = ctx.params
        sw, sb = _sink(pw), _sink(pb)
        dw = db = None
        if not ctx.needs_input_grad[1]:            # frozen prediction head (VALOR.frozen_multimodal): no [vocab x hidden] wgrad
            pass
        elif sw is not None:
            K.gemm(dlog, h, trans_a=True, trans_b=True, out=sw, accumulate=True, defer_done=lambda: _sunk(pw))
        else:
            dw = K.gemm(dlog, h, trans_a=True, trans_b=True)
        if not ctx.needs_input_grad[2]:
            pass
        elif sb is not None:
            K.colsum(dlog, out=sb, accumulate=True); _sunk(pb)
        else:
            db = K.colsum(dlog)
        return dh, dw, db, None, None, None


def decoder_xent_segments(h, w_emb, dec_bias, labels, seg_rows, smoothing=0.0):
    return DecoderXentSegFn.apply(h, w_emb, dec_bias, labels, seg_rows, smoothing)


# ------------------------------------------------------------------------------------------------
class DecoderXentFn(Function):
    """loss = mean CE(h W_emb^T + b, labels): tied-decoder GEMM (modeling.py:253, weight = word embeddings
    :241) + fused softmax cross-entropy (pretrain.py:444). Logits live in a zero-padded [n, Vpad] buffer that
    backward overwrites in place with d(logits)."""

    @staticmethod
    def forward(ctx, h, w_emb, dec_bias, labels, want_logits, smoothing=0.0):
        ctx.smoothing = float(smoothing)
        n, V = h.shape[0], w_emb.shape[0]
        Vpad = (V + 31) // 32 * 32
        buf = _rows_with_slack(n, Vpad, h.dtype, h.device)
        logits = buf[:, :V]
        K.gemm(h, w_emb, bias=dec_bias, out=logits)
        loss_rows = torch.empty(n, dtype=torch.float32, device=h.device)
        lse = torch.empty(n, dtype=torch.float32, device=h.device)
        lib.call("valor_xent_smooth_fwd", _st(), _dt(h), _p(buf), _p(labels), _p(loss_rows), _p(lse), n, V, Vpad, ctx.smoothing)
        loss = torch.empty((), dtype=torch.float32, device=h.device)
        lib.call("valor_mean_f32", _st(), _p(loss_rows), n, _p(loss))
        ctx.save_for_backward(h, w_emb, labels, lse, buf)
        ctx.V = V
        ctx.params = (w_emb, dec_bias)
        if want_logits:
            ctx.mark_non_differentiable(logits)
            return loss, logits
        return loss

    @staticmethod
    def backward(ctx, dloss, *_):
        h, w_emb, labels, lse, buf = ctx.saved_tensors
        n, V, Vpad = h.shape[0], ctx.V, buf.shape[1]
        g = dloss.to(torch.float32).contiguous()
        lib.call("valor_xent_smooth_bwd", _st(), _dt(h), _p(buf), _p(labels), _p(lse), _p(g), 1.0 / n, n, V, Vpad, ctx.smoothing)
        dlog = buf[:, :V]
        dh = K.gemm(dlog, w_emb, trans_b=True)
        pw, pb = ctx.params
        sw, sb = _sink(pw), _sink(pb)
        dw = db = None
        if not ctx.needs_input_grad[1]:            # frozen prediction head (VALOR.frozen_multimodal): no [vocab x hidden] wgrad
            pass
        elif sw is not None:
            K.gemm(dlog, h, trans_a=True, trans_b=True, out=sw, accumulate=True, defer_done=lambda: _sunk(pw))
        else:
            dw = K.gemm(dlog, h, trans_a=True, trans_b=True)
        if not ctx.needs_input_grad[2]:
            pass
        elif sb is not None:
            K.colsum(dlog, out=sb, accumulate=True); _sunk(pb)
        else:
            db = K.colsum(dlog)
        return dh, dw, db, None, None, None


def decoder_xent(h, w_emb, dec_bias, labels, want_logits=False, smoothing=0.0):
    return DecoderXentFn.apply(h, w_emb, dec_bias, labels, want_logits, smoothing)


def decoder_logits(h, w_emb, dec_bias):
    """prediction scores only (compute_loss=False branch, pretrain.py:445-446)."""
    return K.gemm(h, w_emb, bias=dec_bias)


# ------------------------------------------------------------------------------------------------
_FINE_CHUNK = int(os.environ.get("VALOR_FINE_CHUNK", "128"))      # texts per d(sims) tile of the fused backward


class FineContrastFn(Function):
    """loss = InfoNCE(fine_matrix(featA, featB, masks, token weights) * k)  -- pretrain.py:191-211, modeling.py:418-433.
    featA [B,T,D], featB [B,Nv,D] (L2-normalised), wA_raw [B,T], wB_raw [B,Nv] (fp32), maskA/maskB fp32 0/1,
    k = 1/temperature (0-dim fp32 tensor, differentiable).
    bf16 features (the benchmarked arithmetic; every rank runs this on the GATHERED batch): the fused kernels of contrastive_fused.hip --
    the token x token similarities are accumulated and reduced in registers, the backward builds d(sims) for _FINE_CHUNK texts at a
    time; no [B*T, B*Nv] tensor exists. fp32 parity mode (and VALOR_FINE_FUSED=0): S = featA . featB^T through valor_gemm."""

    @staticmethod
    def forward(ctx, featA, featB, wA_raw, wB_raw, maskA, maskB, k):
        B, T, D = featA.shape
        Nv = featB.shape[1]
        dev = featA.device
        f32 = dict(dtype=torch.float32, device=dev)
        fa, fb = featA.contiguous().view(B * T, D), featB.contiguous().view(B * Nv, D)
        fused = featA.dtype == torch.bfloat16 and D % 64 == 0 and lib.load().valor_fine_set_fused(-1) == 1
        ldS = (B * Nv + 7) // 8 * 8
        wA, wB = torch.empty((B, T), **f32), torch.empty((B, Nv), **f32)
        lib.call("valor_fine_weight_softmax", _st(), _p(wA_raw), _p(maskA), _p(wA), B, T)
        lib.call("valor_fine_weight_softmax", _st(), _p(wB_raw), _p(maskB), _p(wB), B, Nv)
        score = torch.empty((B, B), **f32)
        A2B, B2A = torch.empty((B, B, T), **f32), torch.empty((B, B, Nv), **f32)
        idxA = torch.empty((B, B, T), dtype=torch.uint8, device=dev)
        idxB = torch.empty((B, B, Nv), dtype=torch.uint8, device=dev)
        if fused:
            lib.call("valor_fine_fused_fwd", _st(), _p(fa), _p(fb), _p(maskA), _p(maskB), _p(wA), _p(wB), _p(score), _p(A2B), _p(B2A),
                     _p(idxA), _p(idxB), B, B, T, Nv, D)
        else:
            S = torch.empty((B * T, ldS), **f32)
            K.gemm(fa, fb, out=S[:, :B * Nv], out_dtype=torch.float32)
            lib.call("valor_fine_reduce_fwd", _st(), _p(S), ldS, _p(maskA), _p(maskB), _p(wA), _p(wB), _p(score), _p(A2B),
                     _p(B2A), _p(idxA), _p(idxB), B, T, Nv)
        lse_r, lse_c = torch.empty(B, **f32), torch.empty(B, **f32)
        loss = torch.empty((), **f32)
        kk = k.detach().to(torch.float32).contiguous()
        lib.call("valor_infonce_fwd", _st(), _p(score), _p(kk), _p(lse_r), _p(lse_c), _p(loss), B)
        ctx.save_for_backward(fa, fb, maskA, maskB, wA, wB, score, A2B, B2A, idxA, idxB, lse_r, lse_c, kk)
        ctx.dims = (B, T, Nv, D, ldS, featA.dtype, fused)
        ctx.score = score
        return loss

    @staticmethod
    def backward(ctx, dloss):
        fa, fb, maskA, maskB, wA, wB, score, A2B, B2A, idxA, idxB, lse_r, lse_c, kk = ctx.saved_tensors
        B, T, Nv, D, ldS, fdt, fused = ctx.dims
        dev = fa.device
        f32 = dict(dtype=torch.float32, device=dev)
        g = dloss.to(torch.float32).contiguous()
        dscore, dk = torch.empty((B, B), **f32), torch.empty((), **f32)
        part = torch.empty(256, **f32)
        lib.call("valor_infonce_bwd", _st(), _p(score), _p(kk), _p(lse_r), _p(lse_c), _p(g), _p(dscore), _p(dk), _p(part), B)
        dwA, dwB = torch.empty((B, T), **f32), torch.empty((B, Nv), **f32)
        if fused:
            lib.call("valor_fine_weight_grad", _st(), _p(dscore), _p(A2B), _p(B2A), _p(dwA), _p(dwB), B, T, Nv)
            ch = min(B, max(1, _FINE_CHUNK))
            nch = (B + ch - 1) // ch
            dS = torch.empty((ch * T, ldS), dtype=fdt, device=dev)
            if ldS != B * Nv:
                dS[:, B * Nv:].zero_()                       # the leading-dimension padding (whole 16-byte chunks are read)
            dfa = torch.empty((B * T, D), dtype=fdt, device=dev)
            dfb = None
            dfb32 = torch.empty((B * Nv, D), **f32) if nch > 1 else None      # several chunks: their contributions to dfeatB add up in fp32
            for c in range(nch):
                a0 = c * ch
                na = min(ch, B - a0)
                lib.call("valor_fine_ds_chunk", _st(), _dt(dS), _p(dscore), _p(maskA), _p(maskB), _p(wA), _p(wB), _p(idxA), _p(idxB),
                         _p(dS), ldS, a0, na, B, T, Nv)
                dSv = dS[:na * T, :B * Nv]
                K.gemm(dSv, fb, trans_b=True, out=dfa[a0 * T:(a0 + na) * T])                         # dS . featB
                if nch == 1:
                    dfb = K.gemm(dSv, fa, trans_a=True, trans_b=True)                                # dS^T . featA
                else:
                    K.gemm(dSv, fa[a0 * T:(a0 + na) * T], trans_a=True, trans_b=True, out=dfb32, out_dtype=torch.float32, accumulate=c > 0)
            if dfb is None:
                dfb = torch.empty((B * Nv, D), dtype=fdt, device=dev)
                lib.call("valor_cast_from_f32", _st(), _dt(dfb), _p(dfb32), _p(dfb), dfb.numel())
            dfa, dfb = dfa.view(B, T, D), dfb.view(B, Nv, D)
        else:
            dS = torch.zeros((B * T, ldS), dtype=fdt, device=dev)
            lib.call("valor_fine_reduce_bwd", _st(), _dt(dS), _p(dscore), _p(maskA), _p(maskB), _p(wA), _p(wB), _p(A2B), _p(B2A),
                     _p(idxA), _p(idxB), _p(dS), ldS, _p(dwA), _p(dwB), B, T, Nv)
            dSv = dS[:, :B * Nv]
            dfa = K.gemm(dSv, fb, trans_b=True).view(B, T, D)                      # dS . featB
            dfb = K.gemm(dSv, fa, trans_a=True, trans_b=True).view(B, Nv, D)       # dS^T . featA
        dwA_raw, dwB_raw = torch.empty_like(dwA), torch.empty_like(dwB)
        lib.call("valor_fine_weight_softmax_bwd", _st(), _p(wA), _p(dwA), _p(dwA_raw), B, T)
        lib.call("valor_fine_weight_softmax_bwd", _st(), _p(wB), _p(dwB), _p(dwB_raw), B, Nv)
        return dfa, dfb, dwA_raw, dwB_raw, None, None, dk


class FineScoreFn(Function):
    """The fine matrix alone (compute_fine_matrix, pretrain.py:178-211) with UNIT raw token weights, differentiable in the features: what
    late_fusion sums for the tva group (pretrain.py:313-321). Unfused on purpose (S = featA . featB^T through valor_gemm, the reduce kernels
    of contrastive.hip): a rare option, not a benchmarked path."""

    @staticmethod
    def forward(ctx, featA, featB, maskA, maskB):
        B, T, D = featA.shape
        Nv = featB.shape[1]
        dev = featA.device
        f32 = dict(dtype=torch.float32, device=dev)
        fa, fb = featA.contiguous().view(B * T, D), featB.contiguous().view(B * Nv, D)
        ldS = (B * Nv + 7) // 8 * 8
        wA, wB = torch.empty((B, T), **f32), torch.empty((B, Nv), **f32)
        lib.call("valor_fine_weight_softmax", _st(), _p(torch.ones((B, T), **f32)), _p(maskA), _p(wA), B, T)
        lib.call("valor_fine_weight_softmax", _st(), _p(torch.ones((B, Nv), **f32)), _p(maskB), _p(wB), B, Nv)
        score = torch.empty((B, B), **f32)
        A2B, B2A = torch.empty((B, B, T), **f32), torch.empty((B, B, Nv), **f32)
        idxA = torch.empty((B, B, T), dtype=torch.uint8, device=dev)
        idxB = torch.empty((B, B, Nv), dtype=torch.uint8, device=dev)
        S = torch.empty((B * T, ldS), **f32)
        K.gemm(fa, fb, out=S[:, :B * Nv], out_dtype=torch.float32)
        lib.call("valor_fine_reduce_fwd", _st(), _p(S), ldS, _p(maskA), _p(maskB), _p(wA), _p(wB), _p(score), _p(A2B), _p(B2A), _p(idxA), _p(idxB), B, T, Nv)
        ctx.save_for_backward(fa, fb, maskA, maskB, wA, wB, A2B, B2A, idxA, idxB)
        ctx.dims = (B, T, Nv, D, ldS, featA.dtype)
        return score

    @staticmethod
    def backward(ctx, dscore):
        fa, fb, maskA, maskB, wA, wB, A2B, B2A, idxA, idxB = ctx.saved_tensors
        B, T, Nv, D, ldS, fdt = ctx.dims
        dev = fa.device
        f32 = dict(dtype=torch.float32, device=dev)
        dscore = dscore.to(torch.float32).contiguous()
        dwA, dwB = torch.empty((B, T), **f32), torch.empty((B, Nv), **f32)          # gradients of the (constant) token weights: discarded
        dS = torch.zeros((B * T, ldS), dtype=fdt, device=dev)
        lib.call("valor_fine_reduce_bwd", _st(), _dt(dS), _p(dscore), _p(maskA), _p(maskB), _p(wA), _p(wB), _p(A2B), _p(B2A),
                 _p(idxA), _p(idxB), _p(dS), ldS, _p(dwA), _p(dwB), B, T, Nv)
        dSv = dS[:, :B * Nv]
        dfa = K.gemm(dSv, fb, trans_b=True).view(B, T, D)
        dfb = K.gemm(dSv, fa, trans_a=True, trans_b=True).view(B, Nv, D)
        return dfa, dfb, None, None


class InfoNCEFn(Function):
    """contrastive_loss (modeling.py:418-433) of a [B, B] fp32 score matrix with k = 1 / temperature (differentiable)"""

    @staticmethod
    def forward(ctx, score, k):
        B = score.shape[0]
        f32 = dict(dtype=torch.float32, device=score.device)
        score = score.to(torch.float32).contiguous()
        lse_r, lse_c = torch.empty(B, **f32), torch.empty(B, **f32)
        loss = torch.empty((), **f32)
        kk = k.detach().to(torch.float32).contiguous()
        lib.call("valor_infonce_fwd", _st(), _p(score), _p(kk), _p(lse_r), _p(lse_c), _p(loss), B)
        ctx.save_for_backward(score, lse_r, lse_c, kk)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        score, lse_r, lse_c, kk = ctx.saved_tensors
        B = score.shape[0]
        f32 = dict(dtype=torch.float32, device=score.device)
        g = dloss.to(torch.float32).contiguous()
        dscore, dk = torch.empty((B, B), **f32), torch.empty((), **f32)
        part = torch.empty(256, **f32)
        lib.call("valor_infonce_bwd", _st(), _p(score), _p(kk), _p(lse_r), _p(lse_c), _p(g), _p(dscore), _p(dk), _p(part), B)
        return dscore, dk


def late_fusion_fine_contrastive(feat_t, feat_v, feat_a, maskA, k):
    """pretrain.py:313-321: the tva group of contra_type='fine' with late_fusion = InfoNCE(fine(t, v) + fine(t, a)), unit token weights"""
    ones = lambda f: torch.ones(f.shape[:2], dtype=torch.float32, device=f.device)
    s = FineScoreFn.apply(feat_t, feat_v, maskA, ones(feat_v)) + FineScoreFn.apply(feat_t, feat_a, maskA, ones(feat_a))
    return InfoNCEFn.apply(s, k)


class CoarseContrastiveFn(Function):
    """contra_type='coarse' (model/pretrain.py:375-395 + contrastive_loss, modeling.py:418-433): score = featA . (sum_j featB_j)^T over pooled,
    normalised [B, C] features (one B-side operand, or two for the late-fusion sum of the tv and ta matrices), then the symmetric InfoNCE of
    the fine path's kernels. Scores and their gradient stay fp32; the small GEMMs read zero-padded leading dimensions like the fine path."""

    @staticmethod
    def forward(ctx, k, fa, *fbs):
        B, C = fa.shape
        dev = fa.device
        f32 = dict(dtype=torch.float32, device=dev)
        fa = fa.contiguous()
        fbs = tuple(f.contiguous() for f in fbs)
        ldB = (B + 7) // 8 * 8
        S = torch.zeros((B, ldB), **f32)
        for j, fb in enumerate(fbs):
            K.gemm(fa, fb, out=S[:, :B], out_dtype=torch.float32, accumulate=j > 0)
        score = S[:, :B].contiguous()
        lse_r, lse_c = torch.empty(B, **f32), torch.empty(B, **f32)
        loss = torch.empty((), **f32)
        kk = k.detach().to(torch.float32).contiguous()
        lib.call("valor_infonce_fwd", _st(), _p(score), _p(kk), _p(lse_r), _p(lse_c), _p(loss), B)
        ctx.save_for_backward(fa, score, lse_r, lse_c, kk, *fbs)
        ctx.score = score
        return loss

    @staticmethod
    def backward(ctx, dloss):
        fa, score, lse_r, lse_c, kk, *fbs = ctx.saved_tensors
        B, C = fa.shape
        dev, fdt = fa.device, fa.dtype
        f32 = dict(dtype=torch.float32, device=dev)
        g = dloss.to(torch.float32).contiguous()
        dscore, dk = torch.empty((B, B), **f32), torch.empty((), **f32)
        part = torch.empty(256, **f32)
        lib.call("valor_infonce_bwd", _st(), _p(score), _p(kk), _p(lse_r), _p(lse_c), _p(g), _p(dscore), _p(dk), _p(part), B)
        ldB = (B + 7) // 8 * 8
        dS = torch.zeros((B, ldB), dtype=fdt, device=dev)
        dS[:, :B].copy_(dscore)
        dSv = dS[:, :B]
        dfa = None
        for j, fb in enumerate(fbs):
            if dfa is None:
                dfa = K.gemm(dSv, fb, trans_b=True)                                   # dS . featB_j
            else:
                K.gemm(dSv, fb, trans_b=True, out=dfa, accumulate=True)
        dfbs = tuple(K.gemm(dSv, fa, trans_a=True, trans_b=True) for _ in fbs)        # dS^T . featA (the same for every summand)
        return (dk, dfa) + dfbs


def coarse_contrastive(featA, featBs, k):
    return CoarseContrastiveFn.apply(k, featA, *featBs)


def fine_contrastive(featA, featB, wA_raw, wB_raw, maskA, maskB, k):
    return FineContrastFn.apply(featA, featB, wA_raw, wB_raw, maskA, maskB, k)


# ------------------------------------------------------------------------------------------------
class EmbedFn(Function):
    """out[i] = word[ids[i]] + pos[i % L] + typevec   (bert.py:211-215, clip.py:377-379)
    Backward: the tables' gradients go straight into their arena slots (deterministic row sums added to the rows that occur, the
    position sums to the first L rows) -- returned to autograd they cost a zero-filled [vocab, E] temporary and a whole-table add per call."""

    @staticmethod
    def forward(ctx, ids, word, pos, typevec, L):
        n, E = ids.numel(), word.shape[1]
        out = torch.empty((n, E), dtype=word.dtype, device=word.device)
        lib.call("valor_embed_fwd", _st(), _dt(word), _p(ids), _p(word), _p(pos), _p(typevec), _p(out), n, L, E)
        ctx.save_for_backward(ids)
        ctx.cfg = (word.shape, pos.shape if pos is not None else None, typevec is not None, L)
        ctx.params = (word, pos, typevec)
        return out.view(*ids.shape, E)

    @staticmethod
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        wshape, pshape, has_type, L = ctx.cfg
        pw, pp, pt = ctx.params
        d2 = _2d(dout.contiguous())
        n, E = d2.shape
        dword = dpos = dtype_vec = None
        if ctx.needs_input_grad[1]:
            sw = _sink(pw)
            if sw is not None:
                lib.call("valor_embed_bwd_word", _st(), _dt(d2), _p(ids), _p(d2), _p(sw), n, E, 1); _sunk(pw)
            else:
                dword = torch.zeros(wshape, dtype=d2.dtype, device=d2.device)
                lib.call("valor_embed_bwd_word", _st(), _dt(d2), _p(ids), _p(d2), _p(dword), n, E, 0)
        if pshape is not None and ctx.needs_input_grad[2]:
            sp = _sink(pp)
            if sp is not None:
                lib.call("valor_sum_over_batch", _st(), _dt(d2), _p(d2), _p(sp), n // L, L, E, 1); _sunk(pp)
            else:
                dpos = torch.zeros(pshape, dtype=d2.dtype, device=d2.device)
                lib.call("valor_sum_over_batch", _st(), _dt(d2), _p(d2), _p(dpos), n // L, L, E, 0)
        if has_type and ctx.needs_input_grad[3]:
            st_ = _sink(pt)
            if st_ is not None:
                K.colsum(d2, out=st_.view(-1), accumulate=True); _sunk(pt)
            else:
                dtype_vec = K.colsum(d2)
        return None, dword, dpos, dtype_vec, None


def embed(ids, word, pos, typevec, L):
    return EmbedFn.apply(ids.contiguous(), word, pos, typevec, L)


class AssembleFn(Function):
    """tokens = [cls ; patches (+bias)] + pos   (clip.py:264-265 ; modeling.py:755-760). The gradients of cls / pos / bias are added
    straight into their arena slots."""

    @staticmethod
    def forward(ctx, patches, cls, pos, bias, N, Pn):
        E = patches.shape[-1]
        out = torch.empty((N, Pn + 1, E), dtype=patches.dtype, device=patches.device)
        lib.call("valor_assemble_tokens_fwd", _st(), _dt(patches), _p(patches), _p(cls), _p(pos), _p(bias), _p(out), N, Pn, E)
        ctx.cfg = (N, Pn, E, bias is not None, cls.shape, pos.shape)
        ctx.params = (cls, pos, bias)
        return out

    @staticmethod
    def backward(ctx, dout):
        N, Pn, E, has_bias, cshape, pshape = ctx.cfg
        pc, pp, pb = ctx.params
        dout = dout.contiguous()
        dpatch = torch.empty((N * Pn, E), dtype=dout.dtype, device=dout.device)
        sc, sp = _sink(pc), _sink(pp)
        dcls = dposr = dbias = None
        if sc is not None and sp is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            lib.call("valor_assemble_tokens_bwd", _st(), _dt(dout), _p(dout), _p(dpatch), _p(sp), _p(sc), N, Pn, E, 1)
            _sunk(pc); _sunk(pp)
        else:
            dpos = torch.empty((Pn + 1, E), dtype=dout.dtype, device=dout.device)
            lib.call("valor_assemble_tokens_bwd", _st(), _dt(dout), _p(dout), _p(dpatch), _p(dpos), None, N, Pn, E, 0)
            dcls, dposr = dpos[0].clone().view(cshape), dpos.view(pshape)
        if has_bias and ctx.needs_input_grad[3]:
            sb = _sink(pb)
            if sb is not None:
                K.colsum(dpatch, out=sb, accumulate=True); _sunk(pb)
            else:
                dbias = K.colsum(dpatch)
        return dpatch, dcls, dposr, dbias, None, None


def assemble_tokens(patches, cls, pos, bias, N, Pn):
    return AssembleFn.apply(patches, cls, pos, bias, N, Pn)


class PadColsFn(Function):
    """w [N, K] -> zero-padded [N, Kp] copy (data movement only); backward hands the first K gradient columns back."""

    @staticmethod
    def forward(ctx, w, kp):
        ctx.k = w.shape[1]
        out = w.new_zeros((w.shape[0], kp))
        out[:, :w.shape[1]].copy_(w)
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.k].contiguous(), None


def pad_cols(w, kp):
    return PadColsFn.apply(w, kp)


def patchify(images_f32, P, out_dtype, pad_to=0):
    """[N,C,H,W] fp32 -> [N*(H/P)*(W/P), C*P*P] GEMM operand rows in the compute dtype (no grad). pad_to > C*P*P: rows of pad_to
    columns, zero beyond C*P*P (ViT-L/14: 588 -> 592, whole 16-byte chunks for the GEMM's staging)."""
    N, C, H, W = images_f32.shape
    images_f32 = images_f32.contiguous()
    K = C * P * P
    rows = N * (H // P) * (W // P)
    if pad_to > K:
        out = torch.zeros((rows, pad_to), dtype=out_dtype, device=images_f32.device)
    else:
        out = torch.empty((rows, K), dtype=out_dtype, device=images_f32.device)
    lib.call("valor_patchify", _st(), _dt(out), _p(images_f32), _p(out), N, C, H, W, P, out.shape[1])
    return out


class CrossInputFn(Function):
    """va = [video_out + frame_emb + type | audio_out + frame_emb + type]  -> [b, F*X + A*Y, E]
    (modeling.py:485-502 + the torch.cat of bert.py:450)."""

    @staticmethod
    def forward(ctx, vid, aud, vfe, vte, afe, ate):
        b, F, X, E = vid.shape
        _, A, Y, _ = aud.shape
        Sv, Sa = F * X, A * Y
        out = torch.empty((b, Sv + Sa, E), dtype=vid.dtype, device=vid.device)
        lib.call("valor_add_frame_type_fwd", _st(), _dt(vid), _p(vid.contiguous()), _p(vfe), _p(vte), _p(out), b, F, X, E, (Sv + Sa) * E, 0)
        lib.call("valor_add_frame_type_fwd", _st(), _dt(aud), _p(aud.contiguous()), _p(afe), _p(ate), _p(out), b, A, Y, E, (Sv + Sa) * E, Sv)
        ctx.cfg = (b, F, X, A, Y, E, vfe.shape, afe.shape, vte.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        b, F, X, A, Y, E, vfs, afs, ts = ctx.cfg
        dout = dout.contiguous()
        Sv, Sa = F * X, A * Y
        dvid = torch.empty((b, F, X, E), dtype=dout.dtype, device=dout.device)
        daud = torch.empty((b, A, Y, E), dtype=dout.dtype, device=dout.device)
        dvf = torch.zeros(vfs, dtype=dout.dtype, device=dout.device)
        daf = torch.zeros(afs, dtype=dout.dtype, device=dout.device)
        ws = K.workspace(dout.device)
        lib.call("valor_add_frame_type_bwd", _st(), _dt(dout), _p(dout), _p(dvid), _p(dvf), _p(ws), b, F, X, E, (Sv + Sa) * E, 0)
        lib.call("valor_add_frame_type_bwd", _st(), _dt(dout), _p(dout), _p(daud), _p(daf), _p(ws), b, A, Y, E, (Sv + Sa) * E, Sv)
        dvt = K.colsum(dvf.view(-1, E)[:F]).view(ts)
        dat = K.colsum(daf.view(-1, E)[:A]).view(ts)
        return dvid, daud, dvf, dvt, daf, dat


def cross_input(vid, aud, vfe, vte, afe, ate):
    return CrossInputFn.apply(vid, aud, vfe, vte, afe, ate)


class SingleInputFn(Function):
    """one modality only (a dataset without audio or without video): x + frame_emb + type -> [b, F*X, E]
    (modeling.py:485-502; bert.py:452-455 cross-attends to that modality alone)."""

    @staticmethod
    def forward(ctx, x, fe, te):
        b, F, X, E = x.shape
        out = torch.empty((b, F * X, E), dtype=x.dtype, device=x.device)
        lib.call("valor_add_frame_type_fwd", _st(), _dt(x), _p(x.contiguous()), _p(fe), _p(te), _p(out), b, F, X, E, F * X * E, 0)
        ctx.cfg = (b, F, X, E, fe.shape, te.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        b, F, X, E, fs, ts = ctx.cfg
        dout = dout.contiguous()
        dx = torch.empty((b, F, X, E), dtype=dout.dtype, device=dout.device)
        dfe = torch.zeros(fs, dtype=dout.dtype, device=dout.device)
        lib.call("valor_add_frame_type_bwd", _st(), _dt(dout), _p(dout), _p(dx), _p(dfe), _p(K.workspace(dout.device)), b, F, X, E, F * X * E, 0)
        return dx, dfe, K.colsum(dfe.view(-1, E)[:F]).view(ts)


def single_input(x, fe, te):
    return SingleInputFn.apply(x, fe, te)


class L2NormFn(Function):
    @staticmethod
    def forward(ctx, x):
        x2 = _2d(x.contiguous())
        y = torch.empty_like(x2)
        norm = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
        lib.call("valor_l2norm_fwd", _st(), _dt(x2), _p(x2), _p(y), _p(norm), x2.shape[0], x2.shape[1])
        ctx.save_for_backward(y, norm)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        y, norm = ctx.saved_tensors
        d2 = _2d(dy.contiguous())
        dx = torch.empty_like(d2)
        lib.call("valor_l2norm_bwd", _st(), _dt(d2), _p(y), _p(d2), _p(norm), _p(dx), d2.shape[0], d2.shape[1])
        return dx.view(dy.shape)


def l2_normalize(x):
    return L2NormFn.apply(x)


class GatherRowsFn(Function):
    """out[i] = x2d[idx[i]]  (masked-token row selection pretrain.py:441; cls-token pooling modeling.py:387,399)."""

    @staticmethod
    def forward(ctx, x2d, idx):
        x2d = x2d.contiguous()
        n, E = idx.numel(), x2d.shape[1]
        out = torch.empty((n, E), dtype=x2d.dtype, device=x2d.device)
        lib.call("valor_gather_rows", _st(), _dt(x2d), _p(x2d), _p(idx), _p(out), n, E, x2d.stride(0))
        ctx.save_for_backward(idx)
        ctx.shape = x2d.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dout = dout.contiguous()
        dx = torch.zeros(ctx.shape, dtype=dout.dtype, device=dout.device)
        lib.call("valor_scatter_rows", _st(), _dt(dout), _p(dout), _p(idx), _p(dx), idx.numel(), dout.shape[1], dx.stride(0))
        return dx, None


def gather_rows(x2d, idx):
    return GatherRowsFn.apply(x2d, idx)


class TapRowsFn(Function):
    """(x2d, x2d[idx]) as ONE autograd node, for an activation that feeds another consumer besides the row selection (an encoder's
    output: its cls rows go to the contrastive head, modeling.py:387,399, all of it to the decoder's cross-attention input). Backward:
    the rows' gradient is added INTO the other consumer's gradient (gather the few rows, add, write them back) -- a separate
    GatherRowsFn costs a zero-filled [rows, E] tensor plus autograd's add of two [rows, E] tensors. idx must not repeat a row."""

    @staticmethod
    def forward(ctx, x2d, idx):
        x2d = x2d.contiguous()
        ctx.set_materialize_grads(False)
        n, E = idx.numel(), x2d.shape[1]
        out = torch.empty((n, E), dtype=x2d.dtype, device=x2d.device)
        lib.call("valor_gather_rows", _st(), _dt(x2d), _p(x2d), _p(idx), _p(out), n, E, x2d.stride(0))
        ctx.save_for_backward(idx)
        ctx.shape = x2d.shape
        return x2d.view_as(x2d), out

    @staticmethod
    def backward(ctx, dx, drows):
        (idx,) = ctx.saved_tensors
        if drows is None:
            return dx, None
        drows = drows.contiguous()
        n, E = idx.numel(), drows.shape[1]
        if dx is None:
            dx = torch.zeros(ctx.shape, dtype=drows.dtype, device=drows.device)
        else:
            # dx is the other consumer's freshly written gradient (CrossInputFn / SingleInputFn.backward allocate it): updated in place
            dx = dx.contiguous()
            cur = torch.empty_like(drows)
            lib.call("valor_gather_rows", _st(), _dt(dx), _p(dx), _p(idx), _p(cur), n, E, dx.stride(0))
            drows = cur.add_(drows)
        lib.call("valor_scatter_rows", _st(), _dt(drows), _p(drows), _p(idx), _p(dx), n, E, dx.stride(0))
        return dx, None


def tap_rows(x2d, idx):
    """-> (x2d for the other consumer, x2d[idx])"""
    if not FUSE_GLUE:
        return x2d, gather_rows(x2d, idx)
    return TapRowsFn.apply(x2d, idx)


class RowDotFn(Function):
    """y[..., 0] = <x[..., :], w[0, :]> + b   (Linear(E -> 1) of the fine-weight heads, pretrain.py:104-112)"""

    @staticmethod
    def forward(ctx, x, w, b):
        x2 = _2d(x.contiguous())
        y = torch.empty(x2.shape[0], dtype=x.dtype, device=x.device)
        lib.call("valor_rowdot_fwd", _st(), _dt(x2), _p(x2), _p(w), _p(b), _p(y), x2.shape[0], x2.shape[1])
        ctx.save_for_backward(x2, w)
        ctx.cfg = (x.shape, b is not None)
        return y.view(*x.shape[:-1], 1)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        xshape, has_b = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(x2)
        dw = torch.empty_like(w)
        db = torch.empty(1, dtype=w.dtype, device=w.device) if has_b else None
        lib.call("valor_rowdot_bwd", _st(), _dt(x2), _p(dy), _p(x2), _p(w), _p(dx), _p(dw), _p(db), x2.shape[0], x2.shape[1])
        return dx.view(xshape), dw, db


def rowdot(x, w, b):
    return RowDotFn.apply(x, w, b)


# ------------------------------------------------------------------------------------------------ VideoSwin
class WinAttnFn(Function):
    """WindowAttention3D core (videoswin.py:137-160) + roll / window_partition / window_reverse (:205-220) on the fused QKV rows
    in natural token order; the relative position bias table gets its gradient from the kernel's histogram."""

    @staticmethod
    def forward(ctx, qkv, table, geo, n_heads, B):
        qkv = qkv.contiguous()
        o, lse = K.win_attn_fwd(qkv, geo, table, n_heads, B)
        ctx.save_for_backward(qkv, o, lse, table)
        ctx.cfg = (geo, n_heads, B)
        ctx.params = (table,)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse, table = ctx.saved_tensors
        geo, n_heads, B = ctx.cfg
        pt = ctx.params[0]
        st = _sink(pt)
        dqkv, dtable = K.win_attn_bwd(qkv, o, lse, do.contiguous(), geo, table, n_heads, B, dtable=st)
        if st is not None:
            _sunk(pt)
        return dqkv, dtable, None, None, None


def window_attention(qkv, table, geo, n_heads, B):
    return WinAttnFn.apply(qkv, table, geo, n_heads, B)


def patchify3d(video_f32, P, out_dtype):
    """PatchEmbed3D's conv3d input as GEMM rows (videoswin.py:361-369); pixels carry no gradient"""
    return K.patchify3d(video_f32, P, out_dtype)


class GroupMeanFn(Function):
    """mean over X consecutive rows (VideoSwin token pooling for the contrastive head, modeling.py:388-389)"""

    @staticmethod
    def forward(ctx, x2d, X):
        ctx.X = X
        return K.group_mean_fwd(x2d.contiguous(), X)

    @staticmethod
    def backward(ctx, dout):
        return K.group_mean_bwd(dout, ctx.X), None


def group_mean(x2d, X):
    return GroupMeanFn.apply(x2d, X)
