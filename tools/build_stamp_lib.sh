#!/bin/bash
# diagnostic build: libvalor_hip_stamp.so = the library with csrc/gemm8n.hip compiled -DN8_STAMP (per-wave cycle stamps into the split-K
# workspace, read by tools/gemm_stamp.py through VALOR_HIP_LIB). Never the shipped library.
set -e
cd "$(dirname "$0")/.."
python -m valor_amd.build > /dev/null
O=valor_amd/csrc/_obj
objs=$(ls $O/*.o | grep -v "gemm8n.o\|gemm8w.o")
# usage: build_stamp_lib.sh [ablate]: also the ablation builds libvalor_hip_abl{1,2,3}.so (N8_ABLATE: 1 = no DMA, 2 = no fragment reads in the K loop)
[ "$1" = "attn" ] || for v in stamp ${1:+abl1 abl2 abl3 abl4 abl8}; do
  case $v in stamp) D="-DN8_STAMP";; abl1) D="-DN8_STAMP -DN8_ABLATE=1";; abl2) D="-DN8_STAMP -DN8_ABLATE=2";; abl3) D="-DN8_STAMP -DN8_ABLATE=3";; abl4) D="-DN8_STAMP -DN8_ABLATE=4";; abl8) D="-DN8_STAMP -DN8_ABLATE=8";; esac
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $D -c valor_amd/csrc/gemm8n.hip -o /tmp/gemm8n_$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $D -c valor_amd/csrc/gemm8w.hip -o /tmp/gemm8w_$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o valor_amd/libvalor_hip_$v.so $objs /tmp/gemm8n_$v.o /tmp/gemm8w_$v.o && echo valor_amd/libvalor_hip_$v.so ) &
done
wait
# build_stamp_lib.sh attn: libvalor_hip_attstamp.so = the library with csrc/attention_res.hip compiled -DATT_STAMP (tools/attn_stamp.py)
# (ATT_DEFS="-D..." ATT_TAG=_x: extra defines / another file name; ATT_NOSTAMP=" " builds without the stamps: an A/B library for VALOR_HIP_LIB)
if [ "$1" = "attn" ]; then
  objs=$(ls $O/*.o | grep -v "attention_res.o\|attention_xu.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result ${ATT_NOSTAMP:--DATT_STAMP} ${ATT_DEFS} -c valor_amd/csrc/attention_res.hip -o /tmp/attention_res_stamp.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result ${ATT_NOSTAMP:--DATT_STAMP} ${ATT_DEFS} -c valor_amd/csrc/attention_xu.hip -o /tmp/attention_xu_stamp.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o valor_amd/libvalor_hip_attstamp${ATT_TAG}.so $objs /tmp/attention_res_stamp.o /tmp/attention_xu_stamp.o && echo valor_amd/libvalor_hip_attstamp${ATT_TAG}.so
fi
