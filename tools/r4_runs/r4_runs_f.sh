#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4f
for bm in 160 192 256; do timeout -k 10 200 python tools/pp_repeat3.py $bm 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300 | tee -a gpurun_out/r4f/repeat3.txt; done
for cfg in "15 2" "15 1"; do
  set -- $cfg
  MSC_GEMM_PP_MASK=$1 MSC_FUSION_CHAINS=$2 timeout -k 10 200 python tools/determinism_probe.py C3 64 > gpurun_out/r4f/p.txt 2>&1; grep -v amdgpu.ids gpurun_out/r4f/p.txt | tail -4 | tee -a gpurun_out/r4f/probe.txt
done
for rep in 1 2; do
for cfg in "0 2" "1 2" "1 1"; do
  set -- $cfg
  MSC_GEMM_PP=$1 MSC_FUSION_CHAINS=$2 timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > gpurun_out/r4f/bench_pp$1_ch$2_$rep.txt 2>&1
  echo "pp=$1 chains=$2 rep=$rep rc=$? $(tail -1 gpurun_out/r4f/bench_pp$1_ch$2_$rep.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))')" | tee -a gpurun_out/r4f/status.txt
done
done
