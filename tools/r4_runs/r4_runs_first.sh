#!/bin/bash
# round 4: correctness + timing of the ping-pong kernel, the new benchmark-shape tests, the step with / without it
set -o pipefail
mkdir -p gpurun_out/r4a
for k in 160 192 256; do timeout -k 5 120 python tools/pp_debug.py 19584 1536 512 $k > gpurun_out/r4a/pp_debug_$k.txt 2>&1; grep "trial" gpurun_out/r4a/pp_debug_$k.txt | head -3; done
timeout -k 10 600 python tools/pp_check.py > gpurun_out/r4a/pp_check.txt 2>&1; echo "pp_check rc=$?" | tee -a gpurun_out/r4a/status.txt
tail -15 gpurun_out/r4a/pp_check.txt
timeout -k 10 900 python -m pytest tests/test_gemm_shapes_gpu.py -x -q -m gpu > gpurun_out/r4a/shapes_tests.txt 2>&1; echo "shape tests rc=$?" | tee -a gpurun_out/r4a/status.txt
tail -5 gpurun_out/r4a/shapes_tests.txt
for pp in 0 1; do
  MSC_GEMM_PP=$pp timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r4a/bench_pp$pp.txt 2>&1; echo "bench pp=$pp rc=$?" | tee -a gpurun_out/r4a/status.txt
  tail -1 gpurun_out/r4a/bench_pp$pp.txt | cut -c1-400
done
