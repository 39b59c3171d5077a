#!/bin/bash
# quick GPU check: parity tests + bench (no CPU baseline)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-400
