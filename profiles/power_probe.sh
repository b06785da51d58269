#!/bin/bash
# Clocks and power of the GPU while the bench runs (run on the GPU box from the repo root):
#   bash profiles/power_probe.sh [streams]
# Samples rocm-smi twice a second next to `bench.py --steps 5000` (about 20 s); prints the samples and the bench line.
S=${1:-2}
OUT=${OUT:-gpurun_out/power}
mkdir -p $OUT
python bench.py --no-cpu-baseline --streams $S --steps ${STEPS:-5000} --warmup 5 > $OUT/bench_s$S.json 2> $OUT/bench_s$S.err &
BP=$!
sleep 12   # import + build check + warmup
for i in $(seq 30); do
  kill -0 $BP 2>/dev/null || break
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use" | tr '\n' ' '
  echo
  sleep 0.5
done > $OUT/smi_s$S.txt
wait $BP
python -c "import json,sys; r=json.loads(open('$OUT/bench_s$S.json').readlines()[-1]); print('streams', $S, r['value'], 'MP/s', r['ms_per_step'], 'ms/step')"
