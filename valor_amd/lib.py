"""ctypes binding of the C-ABI library ``libvalor_hip.so`` (declared in include/valor_hip.h).

There is NO fallback: if the library is missing or a call returns an error code the
caller gets an exception. Raw device pointers + the current HIP stream are passed; the
kernels never allocate.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VALOR_HIP_LIB: another build of the same library (A/B runs of two kernel versions inside one GPU session, tools/ab_bench.sh)
LIB_PATH = os.environ.get("VALOR_HIP_LIB") or os.path.join(_HERE, "libvalor_hip.so")

DT_BF16 = 0
DT_F32 = 1

ACT_NONE, ACT_GELU_ERF, ACT_QUICK_GELU, ACT_RELU, ACT_TANH = 0, 1, 2, 3, 4
ACT_DERIV = 16      # flag: `preact` / `dact_aux` hold act'(x) instead of x (include/valor_hip.h)

_c = ctypes
_vp, _i, _i64, _u64, _f = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_uint64, _c.c_float

class GemmPolicy(_c.Structure):
    """valor_gemm_policy of include/valor_hip.h: per-call tuning, -1 = the process default. GemmPolicy.make(narrow=1, mfma32=1, ...)"""
    _fields_ = [("key", _i * 12), ("variant", _i), ("tr_asm", _i), ("fast_epilogue", _i), ("sched_256", _i), ("sched_narrow", _i)]
    KEYS = dict(nt_min_k=0, splitk_bf16=1, min_tiles=2, nt_min_tiles=3, raster=4, store=5, nta=6, nn_min_k=7, narrow=8, mfma32=9, wide=10, skinny=11)

    @classmethod
    def make(cls, **kw):
        p = cls()
        for i in range(12):
            p.key[i] = -1
        p.variant = p.tr_asm = p.fast_epilogue = p.sched_256 = p.sched_narrow = -1
        for k, v in kw.items():
            if k in cls.KEYS:
                p.key[cls.KEYS[k]] = int(v)
            else:
                setattr(p, k, int(v))
        return p


class XattnSeg(_c.Structure):
    """valor_xattn_seg of include/valor_hip.h"""
    _fields_ = [("q", _vp), ("o", _vp), ("dout", _vp), ("dq", _vp), ("lse", _vp), ("kv_range", _vp),
                ("q_bs", _i64), ("q_rs", _i64), ("o_bs", _i64), ("o_rs", _i64), ("do_bs", _i64), ("do_rs", _i64), ("dq_bs", _i64), ("dq_rs", _i64),
                ("B", _i), ("Sq", _i), ("seed", _u64), ("offset", _u64)]


# name -> argtypes (restype is always int: 0 ok, <0 error)
SIGNATURES = {
    "valor_gemm": [_vp, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i, _vp, _vp, _i64,
                   _f, _i, _i, _vp, _i64, _vp, _i],
    "valor_gemm_tuned": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i, _vp, _vp, _i64,
                         _f, _i, _i, _vp, _i64, _vp, _i],
    "valor_gemm_kernel_for_tuned": [_vp, _i, _i, _i, _i, _i, _i, _i],
    "valor_gemm_deferred": [_vp, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i, _vp, _vp, _i64,
                            _f, _i, _i, _vp, _i64, _vp, _i, _vp],
    "valor_gemm_pending_bytes": [_vp, _vp],
    "valor_gemm_reduce_group": [_vp, _i, _vp, _i],
    "valor_gemm_set_variant": [_i],
    "valor_gemm_kernel_for": [_i, _i, _i, _i, _i, _i, _i],
    "valor_gemm_set_tr_asm": [_i],
    "valor_gemm_set_fast_epilogue": [_i],
    "valor_gemm_set_policy": [_i, _i],
    "valor_gemm_set_8ph_sched": [_i],
    "valor_gemm_set_narrow_sched": [_i],
    "valor_gemm_narrow_occupancy": [],
    "valor_gemm_wide_occupancy": [],
    "valor_ln_part_blocks": [],
    "valor_ln_set_variant": [_i],
    "valor_ln_set_nt": [_i],
    "valor_bdrln_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _f, _u64, _u64, _vp, _i64, _vp],
    "valor_bdrln_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _u64, _u64, _vp, _i64, _vp],
    "valor_patchify3d": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "valor_group_mean_fwd": [_vp, _i, _vp, _vp, _i64, _i, _i],
    "valor_group_mean_bwd": [_vp, _i, _vp, _vp, _i64, _i, _i],
    "valor_win_attn_workspace_floats": [_i, _i, _i, _i],
    "valor_win_attn_set_variant": [_i],
    "valor_win_attn_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f],
    "valor_win_attn_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _f],
    "valor_colsum_finalize": [_vp, _i, _vp, _i, _i, _vp, _i, _i],
    "valor_colsum_finalize3": [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i],
    "valor_attn_set_variant": [_i],
    "valor_attn_set_res_pipeline": [_i],
    "valor_attn_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                       _vp, _i64, _i64, _vp, _i, _f, _f, _u64, _u64, _vp],
    "valor_attn_decode_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                              _vp, _i64, _i64, _vp, _i64, _f],
    "valor_attn_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i,
                       _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                       _vp, _i64, _i64, _vp, _i, _f, _f, _u64, _u64, _i, _vp],
    "valor_cross_attn_fwd_fused": [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i64, _f, _f, _vp],
    "valor_cross_attn_bwd_fused": [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _f, _vp],
    "valor_reducer_unique_id": [_vp],
    "valor_reducer_create": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i],
    "valor_reducer_launch_bucket": [_vp, _i, _vp, _i],
    "valor_reducer_wait": [_vp, _vp],
    "valor_reducer_destroy": [_vp],
    "valor_xent_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _i64, _i, _i64],
    "valor_xent_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _f, _i64, _i, _i64],
    "valor_xent_smooth_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _i64, _i, _i64, _f],
    "valor_xent_smooth_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _f, _i64, _i, _i64, _f],
    "valor_fine_weight_softmax": [_vp, _vp, _vp, _vp, _i, _i],
    "valor_fine_weight_softmax_bwd": [_vp, _vp, _vp, _vp, _i, _i],
    "valor_fine_reduce_fwd": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i],
    "valor_fine_scores": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i],
    "valor_infonce_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i],
    "valor_infonce_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i],
    "valor_fine_reduce_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i, _i, _i],
    "valor_fine_fused_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i],
    "valor_fine_ds_chunk": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i],
    "valor_fine_weight_grad": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i],
    "valor_fine_set_fused": [_i],
    "valor_adamw_chunk": [],
    "valor_adamw_set_nt": [_i],
    "valor_adamw": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _c.POINTER(_f), _c.POINTER(_f), _i, _f, _f, _f, _i, _i, _vp, _i],
    "valor_grad_norm_clip": [_vp, _i, _vp, _vp, _i64, _f, _f, _vp, _vp, _vp],
    "valor_patchify": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i64],
    "valor_assemble_tokens_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i],
    "valor_assemble_tokens_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i],
    "valor_sum_over_batch": [_vp, _i, _vp, _vp, _i, _i, _i, _i],
    "valor_embed_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i],
    "valor_embed_bwd_word": [_vp, _i, _vp, _vp, _vp, _i64, _i, _i],
    "valor_add_frame_type_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64],
    "valor_add_frame_type_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64],
    "valor_l2norm_fwd": [_vp, _i, _vp, _vp, _vp, _i64, _i],
    "valor_l2norm_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _i64, _i],
    "valor_gather_rows": [_vp, _i, _vp, _vp, _vp, _i64, _i, _i64],
    "valor_decode_prologue": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp],
    "valor_beam_select": [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "valor_scatter_rows": [_vp, _i, _vp, _vp, _vp, _i64, _i, _i64],
    "valor_cast_from_f32": [_vp, _i, _vp, _vp, _i64],
    "valor_dact_mul": [_vp, _i, _vp, _vp, _vp, _i64, _i],
    "valor_mean_f32": [_vp, _vp, _i64, _vp],
    "valor_rowdot_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _i64, _i],
    "valor_rowdot_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i],
    "valor_colsum": [_vp, _i, _vp, _i64, _i, _i64, _vp, _vp, _i, _i],
}

_lib = None


class ValorHipError(RuntimeError):
    pass


def load():
    """Load the library (once). Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ValorHipError(
            f"{LIB_PATH} not found: run `python -m valor_amd.build` (hipcc --offload-arch=gfx950). "
            "valor_amd has no CPU / eager fallback by design.")
    # torch ships its own libamdhip64; it must be the (single) HIP runtime of the process, otherwise
    # streams / device pointers created by torch are foreign to the runtime our kernels launch on.
    import torch  # noqa: F401  (loads torch/lib/libamdhip64.so under its SONAME before our DT_NEEDED resolves)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _i
    _lib = lib
    return lib


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise ValorHipError(f"{name} failed with code {rc}")
    return rc
