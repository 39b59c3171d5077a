#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 300 python tools/host_profile.py > gpurun_out/host.log 2>&1; head -4 gpurun_out/host.log | tail -3; sed -n 9,22p gpurun_out/host.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-200
