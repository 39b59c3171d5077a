#!/bin/bash
# session Q: cross K|V projections on the side stream beside the decoder layers: parity subset + step A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dp_model_gpu.py -q -m gpu -x -k "tiny or north_star or goldens or accumulation or two_ranks or rccl" > gpurun_out/pytest_q.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_q.log
for i in 1 2; do
  VALOR_ENCODER_STREAMS=0 timeout 200 python bench.py --no-cpu-baseline > gpurun_out/ab_q1s_$i.log 2>&1; echo "one stream : $(tail -1 gpurun_out/ab_q1s_$i.log | cut -c50-150)"
  timeout 200 python bench.py --no-cpu-baseline > gpurun_out/ab_q2s_$i.log 2>&1; echo "two streams: $(tail -1 gpurun_out/ab_q2s_$i.log | cut -c50-150)"
done
