#!/bin/bash
# SQ / LDS counter passes for one kernel of one op (run on the GPU box from the repo root):
#   bash profiles/pmc_kernel.sh <run_op.py op> <kernel name substring> [out dir]
# Prints mean counter values per launch; the passes are separate runs (8 SQ slots each), kernel-trace only.
OP=${1:-isp}
KSUB=${2:-wiener_stream}
OUT=${3:-gpurun_out/pmc_$KSUB}
export TMPDIR=/tmp
mkdir -p $OUT
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P3="SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_IFETCH SQ_INSTS_SMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/p$i -o p -- python3 profiles/run_op.py $OP --iters 3 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
meta = None
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            meta = {k: r.get(k) for k in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size', 'Scratch_Size', 'Grid_Size', 'Workgroup_Size')}
dur = []
for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r['Kernel_Name']:
            dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print(ksub, meta, 'mean duration us', sum(dur) / max(len(dur), 1) / 1e3)
for c in sorted(acc):
    print(f'{c:28s} {sum(acc[c]) / len(acc[c]):14.4g}')
PY
