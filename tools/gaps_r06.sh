R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_gaps
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_gaps -o t -- python $R/bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-roofline --no-variants --sim-world 0 > $R/gpurun_out/prof_gaps.log 2>&1; echo "prof rc=$?"
DB=$(find $R/gpurun_out/prof_gaps -name '*.db' | head -1)
python $R/tools/rocpd_gaps.py $DB 580 > $R/gpurun_out/r06_timeline_gaps.txt; cat $R/gpurun_out/r06_timeline_gaps.txt | cut -c1-200
find $R/gpurun_out/prof_gaps -name '*.db' -delete
