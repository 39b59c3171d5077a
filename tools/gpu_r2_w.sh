#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 120 python -m pytest tests/test_finetune_gpu.py -q -m gpu -x -k "tiny" > gpurun_out/pytest_w.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_w.log | cut -c1-300
