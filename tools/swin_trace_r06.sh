#!/bin/bash
# one-stream rocprofv3 kernel trace of the --variant swin step (per-kernel durations), old window kernels (VALOR_WIN_VARIANT=125) and the round-6 default
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for v in 125 7; do
  rm -rf $R/gpurun_out/prof_sw
  VALOR_WIN_VARIANT=$v VALOR_ENCODER_STREAMS=0 VALOR_KV_STREAM=0 timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_sw -o t -- python $R/bench.py --variant swin --steps 5 --warmup 2 --graphs 0 --no-cpu-baseline --no-roofline --no-variants --sim-world 0 > $R/gpurun_out/prof_sw.log 2>&1; echo "prof rc=$?"
  DB=$(find $R/gpurun_out/prof_sw -name '*.db' | head -1)
  python $R/tools/rocpd_stats.py $DB $R/gpurun_out/r06_swin_b64_kernel_stats_win$v.md 40 | head -24 | cut -c1-150
  find $R/gpurun_out/prof_sw -name '*.db' -delete
done
