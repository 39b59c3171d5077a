#!/bin/bash
# diagnostic builds of csrc/attention_win.hip: libvalor_hip_wabl{1,2,4,8,6}.so = the library with WIN_ABLATE set (see attention_win.hip);
# selected through VALOR_HIP_LIB by tools/win_ablate.sh. Never the shipped library.
set -e
cd "$(dirname "$0")/.."
python -m valor_amd.build > /dev/null
O=valor_amd/csrc/_obj
objs=$(ls $O/*.o | grep -v attention_win.o)
for v in 1 2 4 8 12; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -DWIN_ABLATE=$v -c valor_amd/csrc/attention_win.hip -o /tmp/aw_abl$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o valor_amd/libvalor_hip_wabl$v.so $objs /tmp/aw_abl$v.o && echo valor_amd/libvalor_hip_wabl$v.so ) &
done
wait
