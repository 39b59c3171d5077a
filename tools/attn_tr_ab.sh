#!/bin/bash
# A/B of builds of csrc/attention_res.hip (tools/build_stamp_lib.sh attn with ATT_DEFS / ATT_TAG / ATT_NOSTAMP): the pipelined self-attention
# backward at the ViT shape, three interleaved rounds. usage: bash tools/attn_tr_ab.sh lib1.so lib2.so ...
for r in 1 2 3; do
for l in "$@"; do
echo "== $l"; VALOR_HIP_LIB=$l timeout 300 python tools/attn_pipe_ab.py /tmp/x.json 2>&1 | grep "vit_b64 (512" | sed 's/.*pipelined_us.: \([0-9.]*\).*/pipelined_us \1/'
done; done
