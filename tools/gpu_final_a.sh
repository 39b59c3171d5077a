#!/bin/bash
# final round, part A: full GPU parity suite, smoke(), data-parallel path (2 ranks sharing the GPU over gloo) for both variants,
# per-GPU batch sweep of the headline configuration
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for V in clip swin; do
  VALOR_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --variant $V --gpus 2 --steps 3 --warmup 2 --batch 8 --no-cpu-baseline > gpurun_out/dist2_$V.log 2>&1; echo "dist $V rc=$?"; tail -1 gpurun_out/dist2_$V.log | cut -c1-330
done
for B in 32 128; do
  timeout 300 python bench.py --batch $B --no-cpu-baseline > gpurun_out/bench_b$B.log 2>&1; echo "bench b$B rc=$?"; tail -1 gpurun_out/bench_b$B.log | cut -c1-230
done
