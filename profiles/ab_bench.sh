#!/bin/bash
# A/B of two builds of the library on the GPU box with the bench itself (no CPU baseline): in-tree library vs a variant.
# The variant is loaded through TDK_LIB_PATH (torch_darktable/_native.py); the in-tree library is never overwritten.
#   bash profiles/ab_bench.sh <variant.so> [rounds]
V=$1; R=${2:-2}
show() { python -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print(' ', r['value'], 'MP/s', r['ms_per_step'], 'ms/step', {k: round(v*1e3,1) for k,v in r.get('kernel_ms_per_frame',{}).items()})"; }
for i in $(seq $R); do
  for s in 2 1; do
    echo "== in-tree, streams $s"; python bench.py --no-cpu-baseline --streams $s 2>/dev/null | show
    echo "== variant $V, streams $s"; TDK_LIB_PATH=$V python bench.py --no-cpu-baseline --streams $s 2>/dev/null | show
  done
done
