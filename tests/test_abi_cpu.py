"""CPU: the C-ABI library loads, exports every symbol include/valor_hip.h declares, the ctypes table matches the
header, a C translation unit including the header compiles, and the product refuses to run without a GPU
(no CPU fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "valor_hip.h")


def _decls():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\bint\s+(valor_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(2).replace("\n", " ").split(",")]
        out[m.group(1)] = [] if args == ["void"] else args
    return out


def test_library_exports_every_header_symbol():
    from valor_amd import lib
    lib.load()
    so = ctypes.CDLL(lib.LIB_PATH)
    decls = _decls()
    assert len(decls) >= 35
    for name in decls:
        assert hasattr(so, name), f"{name} declared in include/valor_hip.h but not exported"
    assert set(decls) == set(lib.SIGNATURES), set(decls) ^ set(lib.SIGNATURES)
    for name, args in decls.items():
        assert len(args) == len(lib.SIGNATURES[name]), (name, len(args), len(lib.SIGNATURES[name]))
    assert so.valor_adamw_chunk() == 1024 and so.valor_ln_part_blocks() > 0


def test_header_is_plain_c(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "valor_hip.h"\nint (*fp)(void) = valor_adamw_chunk;\nint main(void){ return fp == 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(c), "-o", str(tmp_path / "t.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_argument_validation_without_gpu():
    """entry points validate arguments before touching the device (callable on a CPU-only host)."""
    from valor_amd import lib
    so = lib.load()
    assert so.valor_gemm(None, 0, 0, 0, 4, 4, 8, None, 8, None, 8, None, 4, None, 0, None, None, 0, 1.0, 0, 0, None, 0, None, 0) == -1
    assert so.valor_gemm(None, 0, 0, 0, 0, 4, 8, None, 8, None, 8, None, 4, None, 0, None, None, 0, 1.0, 0, 0, None, 0, None, 0) == 0   # M = 0: no-op
    assert so.valor_bdrln_fwd(None, 0, None, None, None, None, None, None, None, None, None, 4, 768, 1e-5, 0.0, 0, 0, None, 0, None) == -1
    # round-4 entries: the native reducer, the smoothed cross-entropy, the one-launch cross-attention
    import ctypes
    assert so.valor_reducer_unique_id(None) == -1 and so.valor_reducer_destroy(None) == 0
    assert so.valor_reducer_launch_bucket(None, 0, None, 0) == -1 and so.valor_reducer_wait(None, None) == -1
    h = ctypes.c_void_p()
    assert so.valor_reducer_create(ctypes.byref(h), None, 0, 1, 0, None, None, None, 1, 0) == -1
    buf = (ctypes.c_float * 64)()
    lab = (ctypes.c_int64 * 4)()
    assert so.valor_xent_smooth_fwd(None, 1, buf, lab, buf, buf, 4, 16, 16, 1.5) == -1            # smoothing outside [0, 1)
    assert so.valor_xent_smooth_bwd(None, 1, buf, lab, buf, None, 1.0, 4, 1, 16, 0.1) == -1       # smoothing needs V > 1
    assert so.valor_xent_smooth_fwd(None, 1, buf, lab, buf, buf, 0, 16, 16, 0.1) == 0             # no rows: no-op
    assert so.valor_cross_attn_fwd_fused(None, 0, None, 1, buf, buf, 1, 128, 1, 0, 64, 0, 64, 0.125, 0.0, None) == -1      # no segments
    assert so.valor_cross_attn_bwd_fused(None, 0, None, 3, buf, buf, buf, buf, 1, 128, 1, 0, 64, 0, 64, 0, 64, 0, 64, 0.125, 0.0, None) == -1


def test_per_call_gemm_policy_without_touching_process_state():
    """valor_gemm_policy (include/valor_hip.h): the kernel-family choice of ONE call, -1 = the process default. The family query is host
    logic, so the contract is checkable without a GPU: a policy changes the answer for its call only, the process defaults stay."""
    import ctypes
    from valor_amd import lib
    so = lib.load()
    assert ctypes.sizeof(lib.GemmPolicy) == 17 * 4
    M, N, K = 100864, 3072, 768                                    # ViT fc1 forward: family 4 under the default policy
    base = so.valor_gemm_kernel_for(0, 0, 0, M, N, K, 0)
    assert base == 4 and so.valor_gemm_kernel_for_tuned(None, 0, 0, 0, M, N, K, 0) == base
    never = lib.GemmPolicy.make(narrow=0)
    assert so.valor_gemm_kernel_for_tuned(ctypes.addressof(never), 0, 0, 0, M, N, K, 0) == 3          # the 256 x 256 kernel instead
    small = lib.GemmPolicy.make(variant=1)
    assert so.valor_gemm_kernel_for_tuned(ctypes.addressof(small), 0, 0, 0, M, N, K, 0) == 1          # the 128 x 128 kernel pinned
    assert so.valor_gemm_kernel_for(0, 0, 0, M, N, K, 0) == base                                       # nothing global moved
    assert so.valor_gemm_set_policy(8, -1) == 1000 and so.valor_gemm_set_variant(-1) == 4
    # argument validation happens under a policy too
    assert so.valor_gemm_tuned(ctypes.addressof(never), None, 0, 0, 0, 4, 4, 8, None, 8, None, 8, None, 4, None, 0, None, None, 0, 1.0, 0, 0, None, 0,
                               None, 0) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_no_cpu_fallback():
    from valor_amd import kernels as K, lib
    a = torch.zeros(8, 8)
    with pytest.raises(lib.ValorHipError):
        K.gemm(a, a)


def test_integration_snippets_match_header(tmp_path):
    """the C++ binding block compiles against include/valor_hip.h (stub torch headers, -fsyntax-only:
    the compiler checks arity and argument types of every valor_* call) and the ctypes block's argtypes equal the loader's."""
    from valor_amd import lib
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    cpp = re.findall(r"```cpp\n(.*?)```", doc, flags=re.S)
    assert cpp and "valor_bdrln_fwd" in cpp[0]
    src = tmp_path / "snippet.cpp"
    src.write_text(cpp[0])
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused-function", "-I", os.path.join(ROOT, "tests", "stubs"),
                        "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    py = [b for b in re.findall(r"```python\n(.*?)```", doc, flags=re.S) if "argtypes" in b]
    assert py
    m = re.search(r"lib\.(valor_\w+)\.argtypes = \[(.*?)\]", py[0])
    names = dict(vp=ctypes.c_void_p, i64=ctypes.c_int64, f=ctypes.c_float, u64=ctypes.c_uint64, i=ctypes.c_int)
    got = [names[t.strip()] for t in m.group(2).split(",")]
    assert got == lib.SIGNATURES[m.group(1)], (m.group(1), len(got), len(lib.SIGNATURES[m.group(1)]))
    call = re.search(r"lib\.valor_bdrln_fwd\((.*?)\)\s*#", py[0], flags=re.S) or re.search(r"rc = lib\.valor_bdrln_fwd\((.*?)\n\s*if rc", py[0], flags=re.S)
    assert call is not None
