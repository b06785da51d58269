#!/bin/bash
# LDS bank-conflict share and LDS / any-wait shares of every library kernel of a list of ops (run on the GPU box from the repo root):
#   [OPS="ppg bilateral"] [OUT=gpurun_out/pmc_ops] bash profiles/pmc_ops.sh
# One counter pass per op (kernel-trace only); conflict/active = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/pmc_ops}
export OUT
mkdir -p $OUT
for op in ${OPS:-ppg postprocess laplacian bilateral wiener isp}; do
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $OUT/$op -o p -- python3 profiles/run_op.py $op --iters 2 > $OUT/$op.log 2>&1 || echo "$op failed"
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ['OUT']
for op in os.environ.get('OPS', 'ppg postprocess laplacian bilateral wiener isp').split():
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'{out}/{op}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name']
            if 'at::' in k or 'elementwise' in k or 'Functor' in k: continue
            acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==',op)
    for k,c in acc.items():
        m={n: sum(v)/len(v) for n,v in c.items()}
        ia=m.get('SQ_LDS_IDX_ACTIVE',0)
        print(f"  {k:70s} conflict/active={m.get('SQ_LDS_BANK_CONFLICT',0)/ia if ia else 0:.3f} lds_inst={m.get('SQ_INSTS_LDS',0):.3g} valu={m.get('SQ_INSTS_VALU',0):.3g} salu={m.get('SQ_INSTS_SALU',0):.3g} waitlds={m.get('SQ_WAIT_INST_LDS',0)/max(m.get('SQ_WAVE_CYCLES',1),1):.2f} waitany={m.get('SQ_WAIT_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1):.2f}")
PY
