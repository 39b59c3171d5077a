#!/bin/bash
# kernel trace of caption generation (tools/gen_bench.py, greedy): where a decoding step's time goes. usage: tools/gen_trace.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-kvcache}; mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_gen
GEN_MODES=${GEN_MODES:-greedy} timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_gen -o t -- python $R/tools/gen_bench.py $R/gpurun_out/gen_trace_$TAG.json > $R/gpurun_out/prof_gen.log 2>&1; echo "prof rc=$?"
DB=$(find $R/gpurun_out/prof_gen -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB $R/gpurun_out/r06_generation_kernel_stats_$TAG.md 60 | head -50 | cut -c1-170
find $R/gpurun_out/prof_gen -name '*.db' -delete
