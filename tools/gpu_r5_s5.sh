#!/bin/bash
# round 5, session 5: grouped split-K reductions (parity, model goldens, data-parallel model tests, in-step A/B) + the text tower's graph
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_bench_shapes_gpu.py tests/test_graphs_gpu.py -m gpu -q -k "grouped or graph or rng or replay or reducer" > gpurun_out/pytest_s5_a.log 2>&1; echo "a rc=$?"; tail -4 gpurun_out/pytest_s5_a.log
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_dp_model_gpu.py tests/test_finetune_gpu.py -m gpu -q > gpurun_out/pytest_s5_b.log 2>&1; echo "b rc=$?"; tail -4 gpurun_out/pytest_s5_b.log
VALOR_GRAPHS=1 timeout 300 python tools/host_profile.py > gpurun_out/host_graphs1_s5.log 2>&1; echo "host graphs=1 rc=$?"; head -4 gpurun_out/host_graphs1_s5.log | tail -3
STEPS=10 bash tools/step_ab.sh r05_step_ab_s5_group_reduce.txt "per_gemm:VALOR_GROUP_REDUCE=0" "grouped:VALOR_GROUP_REDUCE=1" 2>&1 | tail -4
