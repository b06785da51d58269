#!/bin/bash
# Round-1 measurement capture (run on the GPU box from the repo root):
#   bash profiles/capture_r01.sh
# Writes everything under gpurun_out/r01/; the summaries are then copied into profiles/r01/.
set -o pipefail
OUT=gpurun_out/r01
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 10 --warmup 3 > $OUT/bench_isp_plain.json 2> $OUT/bench_isp_plain.err || exit 1
python3 bench.py --workload rcd --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_rcd_plain.json 2> $OUT/bench_rcd_plain.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_isp_under_rocprof.json 2> $OUT/rocprof_stats.err || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 profiles/run_op.py isp --iters 3 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 profiles/run_op.py isp --iters 3 > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/valu -o v -- python3 profiles/run_op.py isp --iters 3 > $OUT/pmc_valu.log 2>&1 || exit 1
python3 profiles/collect_traffic.py $OUT/fetch $OUT/write $OUT/traffic.json $OUT/valu > $OUT/traffic.log 2>&1 || exit 1
python3 profiles/op_bench.py --storage f16 > $OUT/op_bench_f16.json 2> $OUT/op_bench_f16.err || exit 1
python3 profiles/op_bench.py --storage f32 > $OUT/op_bench_f32.json 2> $OUT/op_bench_f32.err || exit 1
echo capture done
