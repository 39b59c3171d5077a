#!/bin/bash
# session P: encoders on two streams: parity (model, DP, finetune tests) and step A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_dp_model_gpu.py tests/test_finetune_gpu.py -q -m gpu -x > gpurun_out/pytest_p.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_p.log
for i in 1 2; do
  VALOR_ENCODER_STREAMS=0 timeout 200 python bench.py --no-cpu-baseline > gpurun_out/ab_1s_$i.log 2>&1; echo "one stream : $(tail -1 gpurun_out/ab_1s_$i.log | cut -c50-150)"
  timeout 200 python bench.py --no-cpu-baseline > gpurun_out/ab_2s_$i.log 2>&1; echo "two streams: $(tail -1 gpurun_out/ab_2s_$i.log | cut -c50-150)"
done
