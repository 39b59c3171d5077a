#!/bin/bash
# HBM traffic of the dominant GEMM kernels via rocprofv3 PMC passes (separate passes per counter; kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
run() {  # name M N K ta tb
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/pmc/$1_$ctr -o p -- python $R/tools/gemm_one.py $2 $3 $4 $5 $6 3 > $R/gpurun_out/pmc/$1_$ctr.log 2>&1
    DB=$(find $R/gpurun_out/pmc/$1_$ctr -name '*.db' | head -1)
    echo "== $1 $ctr"; python $R/tools/rocpd_pmc.py $DB gemm | grep -v "^dispatch" | sort | uniq -c | head -6
    python $R/tools/rocpd_pmc.py $DB gemm | grep "^dispatch" | head -3
  done
}
cd $R
run wgrad_fc1 3072 768 100864 1 1
run fwd_fc1 100864 3072 768 0 0
run dgrad_fc2 100864 3072 768 0 1
find gpurun_out/pmc -name '*.db' -delete
