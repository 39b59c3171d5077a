#!/bin/bash
# in-session A/B of two builds of libvalor_hip.so on the headline bench (box-to-box variation is ~3 %, within a box ~0.3 %):
#   tools/ab_bench.sh valor_amd/libvalor_hip_prev.so [rounds]      (the other arm is the in-tree library)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
OTHER=$R/$1; N=${2:-2}
for i in $(seq 1 $N); do
  VALOR_HIP_LIB=$OTHER timeout 200 python bench.py --no-cpu-baseline > gpurun_out/ab_other_$i.log 2>&1; echo "other  ($1): $(tail -1 gpurun_out/ab_other_$i.log | cut -c50-150)"
  timeout 200 python bench.py --no-cpu-baseline > gpurun_out/ab_tree_$i.log 2>&1; echo "in-tree            : $(tail -1 gpurun_out/ab_tree_$i.log | cut -c50-150)"
done
