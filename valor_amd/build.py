"""Build libvalor_hip.so (the C-ABI drop-in library) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvalor_hip.so")
OBJDIR = os.path.join(CSRC, "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]


# substrings of (mangled) kernel names that must not touch scratch: the GEMM families, the LDS-resident attention, LayerNorm, AdamW,
# cross-entropy, the fused contrastive forward and the VideoSwin window forward / dK-dV kernels ...
HOT_KERNELS = ("gemm_8ph_kernel", "gemm_8ph2_kernel", "gemm_glds_kernel", "gemm_splitk_reduce", "attn_res_", "attn_x_", "attn_xu_", "ln_fwd", "ln_bwd", "adamw_kernel", "xent_",
               "fine_fused_fwd", "fine_ds_chunk_kernelIDF16bLi16", "win_fwd", "win_bwd_dkv")
# ... except the instantiations whose register demand is known and documented: the key-stationary cross-attention with six
# eight query sub-tiles keeps 128 accumulator registers of O per wave; the backward's dropout variant of four sub-tiles (the six-sub-tile
# backward is gone: attention_xu.hip owns that geometry).
KNOWN_SCRATCH = ("attn_x_fwd_kernelILi8", "attn_x_bwd_kernelILi4ELb1",
                 "kernelIf")          # fp32 (parity mode) instantiations are not in the benchmarked step


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_digest():
    h = hashlib.sha1()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".h"):
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def _compile_one(src, hdig, verbose):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJDIR, src[:-4] + ".o")
    stamp = obj + ".sha1"
    with open(path, "rb") as fh:
        dig = hashlib.sha1(fh.read() + hdig.encode() + " ".join(FLAGS).encode()).hexdigest()
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", obj]
    if verbose:
        print("[valor_amd.build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f"hipcc failed on {src}")
    # a kernel that touches scratch (register arrays demoted to private memory, spills) is a performance bug on this library: the HOT
    # kernels (every kernel of the benchmarked bf16 step except the ones listed in KNOWN_SCRATCH) FAIL the build, the rest is reported.
    fn = None
    bad = []
    for line in r.stderr.splitlines():
        if "Function Name:" in line:
            fn = line.split("Function Name:")[1].split()[0]
        elif "ScratchSize [bytes/lane]:" in line:
            n = int(line.split("ScratchSize [bytes/lane]:")[1].split()[0])
            if n:
                hot = any(h in fn for h in HOT_KERNELS) and not any(k in fn for k in KNOWN_SCRATCH)
                print(f"[valor_amd.build] {'ERROR' if hot else 'WARNING'}: {src}: kernel {fn} uses {n} B/lane of scratch", flush=True)
                if hot:
                    bad.append(fn)
    # strict by default (this library is built by its developers with the pinned ROCm of the image: a hot kernel that starts to spill is a
    # regression to catch at build time); VALOR_BUILD_STRICT=0 downgrades it to the message above for a user on another toolchain, whose
    # register allocation may differ by a few bytes per lane
    if bad and os.environ.get("VALOR_BUILD_STRICT", "1") != "0":
        raise RuntimeError(f"scratch in hot kernels of {src}: {bad} (see -Rpass-analysis=kernel-resource-usage; VALOR_BUILD_STRICT=0 builds anyway)")
    with open(stamp, "w") as fh:
        fh.write(dig)
    return obj, True


def build(verbose=True, force=False):
    os.makedirs(OBJDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    hdig = _headers_digest()
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile_one(s, hdig, verbose), srcs))
    objs = [o for o, _ in res]
    changed = any(c for _, c in res)
    if changed or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[valor_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
