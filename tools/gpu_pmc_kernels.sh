#!/bin/bash
# SQ / TCC counters of the non-GEMM hot kernels (separate rocprofv3 --pmc passes, kernel trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/pmc_kernels; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o p -- python $R/tools/kernels_one.py 1 > $O/trace.log 2>&1; echo "trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $O/sq -o p -- python $R/tools/kernels_one.py 1 > $O/sq.log 2>&1; echo "sq rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p -- python $R/tools/kernels_one.py 1 > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p -- python $R/tools/kernels_one.py 1 > $O/write.log 2>&1; echo "write rc=$?"
cd $R
f() { find $O/$1 -name '*.db' | head -1; }
python tools/pmc_kernels.py gpurun_out/${1:-r04}_pmc_kernels.md $(f trace) $(f sq) $(f fetch) $(f write) 2>&1 | cut -c1-220
find $O -name '*.db' -delete
