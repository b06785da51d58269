set -o pipefail
export TDK_EXTRA_FLAGS=-DTDK_EXPERIMENTS
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "wiener or Wiener" > gpurun_out/ys1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/ys1_tests.log
timeout -k 10 200 python profiles/op_bench.py --only Wiener > gpurun_out/ys1_new.json 2> gpurun_out/ys1_new.err && \
TDK_WIENER_YSTREAM=0 timeout -k 10 200 python profiles/op_bench.py --only Wiener > gpurun_out/ys1_old.json 2> gpurun_out/ys1_old.err
tail -5 gpurun_out/ys1_tests.log
