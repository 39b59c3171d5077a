#!/bin/bash
# session R: decoder weight gradients on the side stream: parity subset + step A/B (VALOR_SIDE_WGRADS=0 keeps them on the main stream)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dp_model_gpu.py -q -m gpu -x -k "tiny or north_star or goldens or two_ranks" > gpurun_out/pytest_r.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_r.log
for i in 1 2; do
  VALOR_SIDE_WGRADS=0 timeout 200 python bench.py --no-cpu-baseline > gpurun_out/ab_rw0_$i.log 2>&1; echo "wgrads main: $(tail -1 gpurun_out/ab_rw0_$i.log | cut -c50-150)"
  timeout 200 python bench.py --no-cpu-baseline > gpurun_out/ab_rw1_$i.log 2>&1; echo "wgrads side: $(tail -1 gpurun_out/ab_rw1_$i.log | cut -c50-150)"
done
