#!/bin/bash
# VideoSwin variant: kernel + model parity tests (and optionally a short bench of the swin configuration)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_swin_gpu.py tests/test_model_gpu.py -q > gpurun_out/pytest_swin.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|error|assert" gpurun_out/pytest_swin.log | head -40
if [ -n "$SWIN_BENCH" ]; then
  timeout 600 python bench.py --variant swin --no-cpu-baseline $SWIN_BENCH > gpurun_out/bench_swin.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench_swin.log | cut -c1-600
fi
