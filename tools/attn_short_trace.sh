R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  rm -rf $R/gpurun_out/prof_short
  VALOR_ATTN_SHORT=$v timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_short -o t -- python $R/tools/attn_short_ab.py /tmp/x.json > /dev/null 2>&1
  DB=$(find $R/gpurun_out/prof_short -name '*.db' | head -1)
  echo "VALOR_ATTN_SHORT=$v"; python $R/tools/rocpd_stats.py $DB /tmp/s.md 20 | grep "attn_" | cut -c1-140
  find $R/gpurun_out/prof_short -name '*.db' -delete
done
