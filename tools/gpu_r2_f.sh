#!/bin/bash
# round 2, session F: attention backward without per-element bounds selects (zero-padded images): parity + A/B timing, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_kernel_variants_gpu.py -m gpu -q > gpurun_out/pytest_attn.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest_attn.log | tail -3
grep -E "^FAILED|^ERROR|^E  " gpurun_out/pytest_attn.log | cut -c1-300 | head -20
timeout 400 python tools/attn_ab.py 0,3 > gpurun_out/attn_ab.log 2>&1; echo "attn rc=$?"; grep -E "FAIL|dropout|vit|ast|dec_self|mlm_self" gpurun_out/attn_ab.log | cut -c1-330
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-200
