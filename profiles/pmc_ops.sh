export TMPDIR=/tmp
mkdir -p gpurun_out/c11
for op in ${OPS:-ppg postprocess laplacian bilateral wiener isp}; do
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d gpurun_out/c11/$op -o p -- python3 profiles/run_op.py $op --iters 2 > gpurun_out/c11/$op.log 2>&1 || echo "$op failed"
done
python3 - <<'PY'
import csv, glob, collections
for op in ['ppg','postprocess','laplacian','bilateral','wiener','isp']:
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'gpurun_out/c11/{op}/**/*counter_collection.csv', recursive=True):
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
