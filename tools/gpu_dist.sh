#!/bin/bash
# smoke() + the data-parallel path on ONE GPU: 2 ranks share the device over gloo (RCCL refuses duplicate GPUs), small batch
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
VALOR_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 2 --batch 8 --no-cpu-baseline > gpurun_out/dist2.log 2>&1; echo "dist rc=$?"; tail -3 gpurun_out/dist2.log | cut -c1-600
timeout 300 python bench.py --steps 3 --warmup 2 --batch 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
