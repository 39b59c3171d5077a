#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_kernel_variants_gpu.py -m gpu -q > gpurun_out/pytest_attn.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/pytest_attn.log | tail -3
timeout 400 python tools/attn_ab.py 0,3 > gpurun_out/attn_ab.log 2>&1; echo "attn rc=$?"; grep -E "FAIL|vit|ast" gpurun_out/attn_ab.log | cut -c1-330
bash tools/ab_bench.sh valor_amd/libvalor_hip_prev.so 2
