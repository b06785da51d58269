#!/bin/bash
# Throughput of the bench (frames on several streams) under the experiment knobs of variants/exp.so (profiles/build_experiments.py):
# which kernels may share a CU decides how much of one frame's chain hides under another's.  Run on the GPU box from the repo root:
#   bash profiles/coresidency_exp.sh "<label>|<ENV=val ...>|<bench args>" ...
LIB=${LIB:-variants/exp.so}
for cfg in "$@"; do
  IFS='|' read -r label envs bargs <<< "$cfg"
  line=$(env TDK_LIB_PATH=$LIB $envs python bench.py --no-cpu-baseline --steps ${STEPS:-200} $bargs 2>/dev/null | tail -1)
  python3 - "$label" "$envs" "$bargs" "$line" <<'PY'
import json, sys
label, envs, bargs, line = sys.argv[1:5]
try:
    r = json.loads(line)
    k = r.get('kernel_ms_per_frame', {})
    print(f"{label:28s} {r['value']:9.1f} MP/s {r['ms_per_step']:7.4f} ms/step  [{envs}] [{bargs}]  alone us: " + ' '.join(f"{n.replace('tdk_', '')}={v * 1e3:.0f}" for n, v in list(k.items())[:4]))
except Exception as e:
    print(f"{label:28s} FAILED ({e}) [{envs}] [{bargs}]")
PY
done
