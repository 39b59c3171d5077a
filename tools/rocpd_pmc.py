"""Per-kernel PMC counter sums from a rocprofv3 (rocpd sqlite) --pmc run.
usage: rocpd_pmc.py results.db [kernel-substring]"""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
c = sqlite3.connect(db)
rows = c.execute("select dispatch_id, kernel_name, counter_name, value, vgpr_count, accum_vgpr_count, lds_block_size, grid_size, workgroup_size "
                 "from counters_collection").fetchall()
per = defaultdict(lambda: defaultdict(float))
meta = {}
for did, kn, cn, val, vg, ag, lds, grid, wg in rows:
    if sub and sub not in kn:
        continue
    per[(did, kn)][cn] += val
    meta[(did, kn)] = (vg, ag, lds, grid, wg)
for (did, kn), cs in sorted(per.items()):
    print(f"dispatch {did} {kn[:80]}  vgpr={meta[(did,kn)][0]} agpr={meta[(did,kn)][1]} lds={meta[(did,kn)][2]} grid={meta[(did,kn)][3]} wg={meta[(did,kn)][4]}")
    for k, v in sorted(cs.items()):
        print(f"    {k:32s} {v:.4g}")
