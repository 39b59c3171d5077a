#!/bin/bash
# session O: QA task + generation tests; Swin bench after the activation / derivative changes
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_finetune_gpu.py -q -m gpu -s > gpurun_out/pytest_o.log 2>&1; echo "pytest rc=$?"; grep -E "bf16 max logit|passed|failed|^FAILED|^E  " gpurun_out/pytest_o.log | head -30
timeout 300 python bench.py --variant swin --no-cpu-baseline > gpurun_out/swin_o.log 2>&1; echo "swin: $(tail -1 gpurun_out/swin_o.log | cut -c1-400)"
