#!/bin/bash
set -o pipefail
O=gpurun_out/r4c; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm_shapes_gpu.py -x -q -m gpu -k "c64" > $O/tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.txt | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/gemm_bench.py > $O/gemm_bench.txt 2>&1; grep "l1 3x3" $O/gemm_bench.txt
MSC_CONV_C64=0 timeout -k 10 300 python tools/gemm_bench.py > $O/gemm_bench_off.txt 2>&1; grep "l1 3x3" $O/gemm_bench_off.txt
for rep in 1 2; do
  for on in 1 0; do
    MSC_CONV_C64=$on timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > $O/bench_${on}_$rep.txt 2>&1
    echo "c64=$on rep=$rep rc=$? $(tail -1 $O/bench_${on}_$rep.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))')" | tee -a $O/status.txt
  done
done
