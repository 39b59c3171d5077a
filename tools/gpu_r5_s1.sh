#!/bin/bash
# round 5, session 1: the new parity cases (16-frame goldens fp32 + bf16, accumulation-window wgrad, fused cross-attention and its fallback,
# data-parallel model tests with the raw-byte token gather) + the default bench line without the CPU baseline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_gemm_bench_shapes_gpu.py tests/test_cross_attn_fused_gpu.py tests/test_dp_model_gpu.py tests/test_attention_gpu.py -m gpu -x -q -k "f16a2 or accumulation_window or cross or dp or attn" > gpurun_out/pytest_s1.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_s1.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_s1.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_s1.log | cut -c1-1500
