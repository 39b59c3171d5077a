#!/bin/bash
# interleaved A/B of hipGraph replay (encoders) against eager issue on the headline bench: tools/graphs_ab.sh OUT [pairs]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/$1; N=${2:-5}; : > $OUT
for i in $(seq 1 $N); do
  for gflag in 0 1; do
    timeout 300 python bench.py --no-cpu-baseline --no-variants --no-roofline --sim-world 0 --steps 20 --warmup 5 --graphs $gflag 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); t = d['timed_region']
print('pair $i graphs=$gflag', d['value'], 'samples/s', d['ms_per_step'], 'ms  launch', t.get('launch_ms_per_step'), 'wait', t.get('wait_ms_per_step'), 'gpu_idle', t.get('gpu_idle_ms_per_step'), 'span', sorted(t['gpu_span_ms'])[len(t['gpu_span_ms'])//2])" | tee -a $OUT
  done
done
python - <<PY | tee -a $OUT
import re, statistics as st
rows = [l.split() for l in open("$OUT") if l.startswith("pair")]
for gflag in ("graphs=0", "graphs=1"):
    v = [float(r[3]) for r in rows if r[2] == gflag]
    print(gflag, "n", len(v), "mean", round(st.mean(v), 2), "median", round(st.median(v), 2), "min", min(v), "max", max(v), "stdev", round(st.pstdev(v), 2))
d = [float(b[3]) - float(a[3]) for a, b in zip(rows[0::2], rows[1::2])]
print("paired differences graphs - eager:", [round(x, 2) for x in d], "mean", round(st.mean(d), 2), "wins", sum(x > 0 for x in d), "of", len(d))
PY
