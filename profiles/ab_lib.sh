#!/bin/bash
# A/B of two builds of the library on the GPU box: op_bench rows for the given ops with the in-tree library, then with a
# variant loaded through TDK_LIB_PATH (the in-tree library is never overwritten).
#   bash profiles/ab_lib.sh <variant.so> "<op filter>" [storage]
V=$1; F=$2; S=${3:-f16}
rows() { python -c "import sys,json; [print(' ', r['op'], r['ms'], {k: round(v,4) for k,v in r.get('kernels',{}).items()}) for r in map(json.loads, sys.stdin) if 'ms' in r]"; }
for round in 1 2; do
  echo "== in-tree"; python profiles/op_bench.py --storage $S --only "$F" 2>/dev/null | rows
  echo "== variant $V"; TDK_LIB_PATH=$V python profiles/op_bench.py --storage $S --only "$F" 2>/dev/null | rows
done
