#!/bin/bash
mkdir -p gpurun_out/r4s
run() { env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3))'; }
for rep in 1 2; do
  for wb in 384 512 640 768 1024; do echo "rep $rep MSC_WGRAD_BLOCKS=$wb $(run MSC_WGRAD_BLOCKS=$wb)" | tee -a gpurun_out/r4s/sweep.txt; done
  echo "rep $rep MSC_SIDE_PRIO=0 $(run MSC_SIDE_PRIO=0)" | tee -a gpurun_out/r4s/sweep.txt
  echo "rep $rep MSC_GEMM_PP=192 $(run MSC_GEMM_PP=192)" | tee -a gpurun_out/r4s/sweep.txt
  echo "rep $rep MSC_GEMM_PP_MASK=11 (fc1 fwd tiled) $(run MSC_GEMM_PP_MASK=11)" | tee -a gpurun_out/r4s/sweep.txt
done
