#!/bin/bash
# session L: host-side cost of a step; Swin variant old vs new activations
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 300 python tools/host_profile.py > gpurun_out/host.log 2>&1; head -4 gpurun_out/host.log | tail -3; sed -n 9,30p gpurun_out/host.log
for i in 1 2; do
VALOR_HIP_LIB=$PWD/valor_amd/libvalor_hip_prev.so timeout 300 python bench.py --variant swin --no-cpu-baseline > gpurun_out/swin_prev_$i.log 2>&1; echo "swin prev: $(tail -1 gpurun_out/swin_prev_$i.log | cut -c50-160)"
timeout 300 python bench.py --variant swin --no-cpu-baseline > gpurun_out/swin_tree_$i.log 2>&1; echo "swin tree: $(tail -1 gpurun_out/swin_tree_$i.log | cut -c50-160)"
done
