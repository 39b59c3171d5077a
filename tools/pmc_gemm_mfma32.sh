#!/bin/bash
# SQ counters of ONE family-4 forward GEMM under the two MFMA shapes (VALOR_GEMM_MFMA32 = 0: 16x16x32, 1: 32x32x16), separate rocprofv3
# --pmc passes (kernel trace only). usage: tools/pmc_gemm_mfma32.sh OUTTAG M N K
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/pmc_gemm_$TAG; mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
P3="GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_MFMA"
for m in 0 1; do
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    VALOR_GEMM_NARROW=1 VALOR_GEMM_MFMA32=$m timeout 200 rocprofv3 --kernel-trace --pmc $P -d $O/n${m}_p$i -o p --output-format csv -- python $R/tools/gemm_one.py "$@" 0 0 4 > $O/n${m}_p$i.log 2>&1; echo "mfma32=$m pass$i rc=$?"
  done
done
cd $R
python tools/pmc_gemm_sq.py $O | sed 's/n0 = base, n1 = narrow family 4/n0 = 16x16x32, n1 = 32x32x16/' > gpurun_out/pmc_gemm_$TAG.md; cat gpurun_out/pmc_gemm_$TAG.md | cut -c1-250
find $O -name '*.db' -delete
