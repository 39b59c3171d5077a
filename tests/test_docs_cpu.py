"""The documents cite evidence by path: every `profiles/...`, `tools/...`, `tests/...`, `oracle/...`, `valor_amd/...` file they name exists,
and the activation ids of the C header and of the ctypes table agree."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "INTEGRATION.md", "README.md"]


def _expand(path):
    """`a_{x,y}_b.json` -> both; `r02_*` / `r0N_...` style wildcards -> glob"""
    m = re.search(r"\{([^{}]*)\}", path)
    if m:
        out = []
        for alt in m.group(1).split(","):
            out += _expand(path[:m.start()] + alt + path[m.end():])
        return out
    return [path]


def test_cited_files_exist():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for raw in set(re.findall(r"`((?:profiles|tools|tests|oracle|valor_amd|include)/[A-Za-z0-9_./{},*\-]+)`", text)):
            raw = raw.rstrip(".,")
            if "…" in raw or raw.endswith("/"):
                continue
            for path in _expand(raw):
                path = path.split("::")[0]
                if "*" in path or "N" in os.path.basename(path).split("_")[0]:      # r02_*.json, r0N_... : a family of files
                    if not glob.glob(os.path.join(ROOT, path.replace("r0N", "r0*"))):
                        missing.append((doc, raw))
                elif not os.path.exists(os.path.join(ROOT, path)):
                    missing.append((doc, raw))
    # bare profile names (`r02_x.json` inside a sentence that already said profiles/)
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for raw in set(re.findall(r"`(r0\d_[A-Za-z0-9_.{},*\-]+\.(?:json|txt|md))`", text)):
            if not any(glob.glob(os.path.join(ROOT, "profiles", x)) for x in _expand(raw)):
                missing.append((doc, raw))
    assert not missing, missing


def test_activation_ids_match_the_header():
    from valor_amd import lib
    hdr = open(os.path.join(ROOT, "include", "valor_hip.h")).read()
    ids = {k: int(v) for k, v in re.findall(r"#define VALOR_ACT_(\w+)\s+(\d+)", hdr)}
    assert ids == {"NONE": lib.ACT_NONE, "GELU_ERF": lib.ACT_GELU_ERF, "QUICK_GELU": lib.ACT_QUICK_GELU, "RELU": lib.ACT_RELU,
                   "TANH": lib.ACT_TANH, "DERIV": lib.ACT_DERIV}
    common = open(os.path.join(ROOT, "valor_amd", "csrc", "common.h")).read()
    cids = {k: int(v) for k, v in re.findall(r"#define VALOR_ACT_(\w+)\s+(\d+)", common)}
    for k, v in ids.items():
        assert cids[k] == v, k
