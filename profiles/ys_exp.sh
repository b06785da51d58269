set -o pipefail
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "wiener or Wiener or fusion or image_processor or process_image_set" > gpurun_out/ys3_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/ys3_tests.log
tail -4 gpurun_out/ys3_tests.log
python profiles/wiener_ablate_exp.py variants/ys_full.so variants/ys_timing.so > gpurun_out/ys_timing8.txt 2>&1; cat gpurun_out/ys_timing8.txt
