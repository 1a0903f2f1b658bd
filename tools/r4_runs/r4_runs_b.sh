#!/bin/bash
# round 4, run b: the step with the ping-pong kernel (two fusion chains / one chain) against the tiled kernel, same box; new tests
set -o pipefail
mkdir -p gpurun_out/r4b
timeout -k 10 900 python -m pytest tests/test_gemm_shapes_gpu.py -x -q -m gpu > gpurun_out/r4b/shapes_tests.txt 2>&1; echo "shape tests rc=$?" | tee -a gpurun_out/r4b/status.txt
tail -3 gpurun_out/r4b/shapes_tests.txt
for rep in 1 2; do
for cfg in "0 2" "1 2" "1 1"; do
  set -- $cfg
  MSC_GEMM_PP=$1 MSC_FUSION_CHAINS=$2 timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > gpurun_out/r4b/bench_pp$1_ch$2_$rep.txt 2>&1
  echo "pp=$1 chains=$2 rep=$rep rc=$? $(tail -1 gpurun_out/r4b/bench_pp$1_ch$2_$rep.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],3), round(j["loss"],4))')" | tee -a gpurun_out/r4b/status.txt
done
done
MSC_GEMM_PP=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r4b/bench_full.txt 2>&1; echo "full bench rc=$?" | tee -a gpurun_out/r4b/status.txt
tail -1 gpurun_out/r4b/bench_full.txt | cut -c1-300
