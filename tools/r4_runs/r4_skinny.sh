#!/bin/bash
set -o pipefail
O=gpurun_out/r4s; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm_shapes_gpu.py -x -q -m gpu -k "skinny" > $O/tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.txt | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/skinny_bench.py > $O/skinny_bench.txt 2>&1; cat $O/skinny_bench.txt | grep -v amdgpu.ids | cut -c1-250
for rep in 1 2; do
  for sk in 1 0; do
    MSC_GEMM_SKINNY=$sk timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > $O/bench_${sk}_$rep.txt 2>&1
    echo "skinny=$sk rep=$rep rc=$? $(tail -1 $O/bench_${sk}_$rep.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))')" | tee -a $O/status.txt
  done
done
