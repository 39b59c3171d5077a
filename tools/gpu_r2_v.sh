#!/bin/bash
# last session of round 2: the other configurations on the final build
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 120 python bench.py --variant swin --no-cpu-baseline > gpurun_out/final_swin.log 2>&1; echo "swin: $(tail -1 gpurun_out/final_swin.log | cut -c1-170)"
timeout 150 python bench.py --variant large --no-cpu-baseline --steps 4 > gpurun_out/final_large.log 2>&1; echo "large: $(tail -1 gpurun_out/final_large.log | cut -c1-170)"
