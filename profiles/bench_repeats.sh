#!/bin/bash
# The bench line of record + how much it moves from run to run on one box (run on the GPU box from the repo root):
#   bash profiles/bench_repeats.sh <tag> [n]
# Writes gpurun_out/<tag>/bench_isp_plain.json (one full default run: CPU baseline + parity check) and bench_isp_repeats.txt
# (n more runs of `bench.py --steps 20 --warmup 5 --no-cpu-baseline`, the driver's step counts, at 3 and at 2 streams).
TAG=${1:-r04}; N=${2:-5}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_isp_plain.json 2> $OUT/bench_isp_plain.err || exit 1
for s in 3 2; do
  for i in $(seq $N); do
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams $s 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.readlines()[-1]); print('streams $s', r['value'], 'MP/s', r['ms_per_step'], 'ms/step')"
  done
done > $OUT/bench_isp_repeats.txt
python3 -c "import json; r=json.load(open('$OUT/bench_isp_plain.json')); print('full run', r['value'], r['ms_per_step'], r['parity_check']['ok'])" >> $OUT/bench_isp_repeats.txt
cat $OUT/bench_isp_repeats.txt
