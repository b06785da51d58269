#!/bin/bash
# A/B of two builds of the library on the GPU box: op_bench rows for the given ops with the in-tree library, then with
# a variant copied over it (the box is a scratch copy of the tree).   bash profiles/ab_lib.sh <variant.so> "<op filter>" [storage]
V=$1; F=$2; S=${3:-f16}
LIB=torch-darktable_amd/torch_darktable/libtdk_hip.so
cp $LIB /tmp/tdk_new.so
for round in 1 2; do
  echo "== new"; cp /tmp/tdk_new.so $LIB; python profiles/op_bench.py --storage $S --only "$F" 2>/dev/null | python -c "import sys,json; [print(' ', r['op'], r['ms'], {k: round(v,4) for k,v in r.get('kernels',{}).items()}) for r in map(json.loads, sys.stdin) if 'ms' in r]"
  echo "== variant $V"; cp $V $LIB; python profiles/op_bench.py --storage $S --only "$F" 2>/dev/null | python -c "import sys,json; [print(' ', r['op'], r['ms'], {k: round(v,4) for k,v in r.get('kernels',{}).items()}) for r in map(json.loads, sys.stdin) if 'ms' in r]"
done
cp /tmp/tdk_new.so $LIB
