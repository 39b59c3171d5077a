#!/bin/bash
# SQ counters of the window-attention kernels (one rocprofv3 --pmc pass, kernel trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/pmc_win
python tools/win_one.py 2 3 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_win/sq -o p -- python $R/tools/win_one.py 2 1 > $R/gpurun_out/pmc_win/sq.log 2>&1; echo "pmc rc=$?"
cd $R
DB=$(find gpurun_out/pmc_win/sq -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_pmc.py $DB win_ > gpurun_out/pmc_win/sq_counters.txt; head -60 gpurun_out/pmc_win/sq_counters.txt
find gpurun_out/pmc_win -name '*.db' -delete
