#!/bin/bash
# Measurement capture (run on the GPU box from the repo root):
#   bash profiles/capture.sh <tag> <git-short-hash>
# Writes everything under gpurun_out/<tag>/; the summaries are then copied into profiles/<tag>/.
# PMC passes are separate runs with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# SQ has 8 slots per pass).
set -o pipefail
TAG=${1:-r02}
GIT=${2:-unknown}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --steps 10 --warmup 3 > $OUT/bench_isp_plain.json 2> $OUT/bench_isp_plain.err || exit 1
python3 bench.py --workload rcd --steps 20 --warmup 5 > $OUT/bench_rcd_plain.json 2> $OUT/bench_rcd_plain.err || exit 1
python3 bench.py --workload ppg_wiener50 --steps 5 --warmup 2 > $OUT/bench_ppg_wiener50_plain.json 2> $OUT/bench_ppg_wiener50_plain.err || exit 1
python3 bench.py --steps 10 --warmup 3 --streams 1 --no-cpu-baseline > $OUT/bench_isp_streams1_plain.json 2> $OUT/bench_isp_streams1_plain.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_isp_under_rocprof.json 2> $OUT/rocprof_stats.err || exit 1
# the same command with the frames back to back on one stream: per-kernel durations with the GPU to themselves
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o bench -- python3 bench.py --steps 5 --warmup 2 --streams 1 --no-cpu-baseline > $OUT/bench_isp_streams1_under_rocprof.json 2> $OUT/rocprof_stats1.err || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 profiles/run_op.py isp_jpeg --iters 3 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 profiles/run_op.py isp_jpeg --iters 3 > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/valu -o v -- python3 profiles/run_op.py isp_jpeg --iters 3 > $OUT/pmc_valu.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -o s -- python3 profiles/run_op.py isp_jpeg --iters 3 > $OUT/pmc_sq2.log 2>&1 || echo "sq2 pass failed (optional)"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/sq3 -o t -- python3 profiles/run_op.py isp_jpeg --iters 3 > $OUT/pmc_sq3.log 2>&1 || echo "sq3 pass failed (optional)"
python3 profiles/collect_traffic.py $OUT/fetch $OUT/write $OUT/traffic.json $OUT/valu $GIT $OUT/sq2 $OUT/sq3 > $OUT/traffic.log 2>&1 || exit 1
# the bench line again, now that the counters of these sources exist (roofline.valu / roofline.composite need them)
cp $OUT/traffic.json profiles/traffic.json
python3 bench.py --steps 10 --warmup 3 > $OUT/bench_isp_plain.json 2> $OUT/bench_isp_plain.err || exit 1
python3 bench.py --steps 10 --warmup 3 --streams 2 --no-cpu-baseline > $OUT/bench_isp_streams2_plain.json 2> $OUT/bench_isp_streams2_plain.err || echo "streams 2 failed"
python3 profiles/op_bench.py --storage f16 > $OUT/op_bench_f16.json 2> $OUT/op_bench_f16.err || echo "op bench f16 failed"
python3 profiles/op_bench.py --storage f32 > $OUT/op_bench_f32.json 2> $OUT/op_bench_f32.err || echo "op bench f32 failed"
python3 profiles/op_bench.py --storage f16 --width 8192 --height 6144 --only "PPG|Wiener.process C=3" > $OUT/op_bench_50mp_f16.json 2> $OUT/op_bench_50mp_f16.err || echo "op bench 50 MP failed"
python3 profiles/jpeg_bench.py --pillow > $OUT/jpeg_bench.txt 2>&1 || echo "jpeg bench failed"
# rocprofv3 kernel table of the same encodes (its averages against the event-timer figures of jpeg_bench.txt)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_jpeg -o jpeg -- python3 profiles/jpeg_bench.py --iters 10 > $OUT/jpeg_under_rocprof.txt 2>&1 || echo "jpeg rocprof failed"
python3 profiles/laplacian_kernels.py > $OUT/laplacian_kernels.txt 2>&1 || echo "laplacian kernels failed"
echo capture done
