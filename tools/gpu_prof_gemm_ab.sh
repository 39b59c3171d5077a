#!/bin/bash
# one-stream kernel traces of the bench step under two GEMM policies, GEMM kernels grouped by grid (= shape)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
export VALOR_ENCODER_STREAMS=0 VALOR_KV_STREAM=0
for tag in new old; do
  if [ $tag = old ]; then export VALOR_GEMM_NARROW=0 VALOR_GEMM_8PH_SCHED=0; fi
  rm -rf $R/gpurun_out/prof_$tag
  timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_$tag -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_$tag.log 2>&1; echo "prof $tag rc=$?"
  DB=$(find $R/gpurun_out/prof_$tag -name '*.db' | head -1)
  python $R/tools/rocpd_gemm_by_grid.py $DB 6 > $R/gpurun_out/r4_gemm_by_grid_$tag.txt
  python $R/tools/rocpd_stats.py $DB $R/gpurun_out/r4_kernel_stats_$tag.md 60 > /dev/null
  find $R/gpurun_out/prof_$tag -name '*.db' -delete
done
head -12 $R/gpurun_out/r4_gemm_by_grid_new.txt; head -12 $R/gpurun_out/r4_gemm_by_grid_old.txt
