"""TEST INFRASTRUCTURE: copy the MODEL / OPTIMIZER part of the reference's shipped configuration files into tests/golden/ (the GPU
box has no /root/reference): every top-level key except the dataset paths; data_cfg keeps only task / batch_size / sample counts.
The text is re-serialised WITHOUT its final closing brace, exactly the defect the shipped files have (they do not json.load), so
valor_amd.config.parse_json_lenient is exercised on the committed fixture too.   python oracle/make_config_fixtures.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from valor_amd.config import parse_json_lenient  # noqa: E402

for name in ("base", "large"):
    src = f"/root/reference/config/pretrain-VALOR-{name}.json"
    text = open(src).read()
    try:
        json.loads(text); shipped_ok = True
    except json.JSONDecodeError:
        shipped_ok = False
    cfg = parse_json_lenient(text)
    keep = ("task", "batch_size", "video_sample_num", "audio_sample_num", "max_txt_len", "epoch", "n_workers")
    cfg["data_cfg"] = {"train": [{k: d[k] for k in keep if k in d} for d in cfg["data_cfg"]["train"]], "val": []}
    out = json.dumps(cfg, indent=1)
    if not shipped_ok:
        out = out.rstrip()[:-1].rstrip()          # drop the last brace again: the fixture is as broken as the original
    path = os.path.join(ROOT, "tests", "golden", f"pretrain-VALOR-{name}.json")
    open(path, "w").write(out + "\n")
    print(path, "shipped file parses:", shipped_ok, "tasks:", [d["task"] for d in cfg["data_cfg"]["train"]][:2])
