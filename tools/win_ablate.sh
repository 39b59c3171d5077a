#!/bin/bash
# where the window-attention kernels spend their time: tools/win_one.py (VideoSwin-B stage $1 at b = 64, 8 frames) under a kernel trace with the
# shipped library and with the WIN_ABLATE builds of tools/build_win_ablate.sh (1 = no table fill, 2 = no tile loop, 4 = bias gather reads entry 0,
# 8 = no exponential, 12 = both). Prints avg us per kernel and build.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
ST=${1:-2}
for v in base 1 2 4 8 12; do
  L=$R/valor_amd/libvalor_hip.so; [ $v != base ] && L=$R/valor_amd/libvalor_hip_wabl$v.so
  rm -rf /tmp/wabl_$v
  VALOR_HIP_LIB=$L timeout 200 rocprofv3 --kernel-trace -d /tmp/wabl_$v -o t -- python $R/tools/win_one.py $ST 3 > /tmp/wabl_$v.log 2>&1
  DB=$(find /tmp/wabl_$v -name '*.db' | head -1)
  echo "== build $v"; python $R/tools/rocpd_stats.py $DB /tmp/wabl_$v.md 12 | grep -E "win_" | cut -c1-140
done
