#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_graphs_gpu.py -m gpu -q > gpurun_out/pytest_s4_graphs.log 2>&1; echo "graphs pytest rc=$?"; grep -n "names differ\|passed\|failed\|Error" gpurun_out/pytest_s4_graphs.log | cut -c1-1500 | head -20
for g in 0 1; do VALOR_GRAPHS=$g timeout 300 python tools/host_profile.py > gpurun_out/host_graphs$g.log 2>&1; echo "host graphs=$g rc=$?"; head -4 gpurun_out/host_graphs$g.log | tail -3; done
STEPS=10 bash tools/step_ab.sh r05_step_ab_s4_graphs.txt "eager:VALOR_GRAPHS=0" "graphs:VALOR_GRAPHS=1" 2>&1 | tail -6
