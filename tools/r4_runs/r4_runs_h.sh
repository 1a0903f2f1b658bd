#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4h
timeout -k 10 900 python -m pytest tests/test_gemm_shapes_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "linear or ping_pong or gelu or determinism" > gpurun_out/r4h/tests.txt 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r4h/status.txt
tail -4 gpurun_out/r4h/tests.txt | cut -c1-250
timeout -k 10 600 python tools/pp_check.py 19584 3 > gpurun_out/r4h/pp_check.txt 2>&1; grep -E "fwd|dgrd|TOTAL|8192|4096x" gpurun_out/r4h/pp_check.txt | cut -c1-330
for rep in 1 2; do
  timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > gpurun_out/r4h/bench_$rep.txt 2>&1
  echo "rep=$rep rc=$? $(tail -1 gpurun_out/r4h/bench_$rep.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))')" | tee -a gpurun_out/r4h/status.txt
done
MSC_GEMM_PP=0 timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print("pp=0:", round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))' | tee -a gpurun_out/r4h/status.txt
