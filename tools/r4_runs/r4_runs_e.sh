#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4e
timeout -k 10 200 python tools/pp_repeat.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4e/repeat.txt
for cfg in "1 1 0" "1 2 512" "1 2 1536" "1 2 2048" "1 1 512" "1 1 2048"; do
  set -- $cfg
  MSC_GEMM_PP_MASK=$1 MSC_FUSION_CHAINS=$2 MSC_GEMM_PP_K=$3 timeout -k 10 200 python tools/determinism_probe.py C3 64 2>&1 | grep -v amdgpu.ids | head -4 | tee -a gpurun_out/r4e/probe.txt
done
