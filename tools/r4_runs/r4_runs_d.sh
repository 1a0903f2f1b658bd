#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4d
for mask in 15 0 1 2 4 8; do
  MSC_GEMM_PP_MASK=$mask timeout -k 10 200 python tools/determinism_probe.py C3 64 2>&1 | grep -v amdgpu.ids | head -6 | tee -a gpurun_out/r4d/probe.txt
done
for mask in 15 1 2 4 8; do
for rep in 1 2; do
  MSC_GEMM_PP_MASK=$mask timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > gpurun_out/r4d/bench_m${mask}_$rep.txt 2>&1
  echo "mask=$mask rep=$rep rc=$? $(tail -1 gpurun_out/r4d/bench_m${mask}_$rep.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))')" | tee -a gpurun_out/r4d/status.txt
done
done
