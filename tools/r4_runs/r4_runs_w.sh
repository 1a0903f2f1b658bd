#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4w
timeout -k 10 600 python -m pytest tests/test_gemm_shapes_gpu.py -x -q -m gpu -k "wgrad" > gpurun_out/r4w/wgrad_tests.txt 2>&1; echo "wgrad tests rc=$?" | tee -a gpurun_out/r4w/status.txt
tail -12 gpurun_out/r4w/wgrad_tests.txt | cut -c1-250
timeout -k 10 600 python tools/pp_check.py 19584 3 > gpurun_out/r4w/pp_check.txt 2>&1; grep -i "wgrad" gpurun_out/r4w/pp_check.txt
for rep in 1 2; do for v in 0 1; do
  MSC_GEMM_PPW=$v timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > gpurun_out/r4w/bench_ppw$v_$rep.txt 2>&1
  echo "ppw=$v rep=$rep rc=$? $(tail -1 gpurun_out/r4w/bench_ppw$v_$rep.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))')" | tee -a gpurun_out/r4w/status.txt
done; done
