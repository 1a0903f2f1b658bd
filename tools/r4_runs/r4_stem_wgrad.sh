#!/bin/bash
# stem weight gradient on conv_stem.hip (MSC_STEM_WGRAD_HALO=1) vs the gathered split-K GEMM, and where the encoder half of the optimizer update
# is enqueued (MSC_STEM_HOOK_AT: before the stem's BatchNorm / pooling backward or before its weight gradient; MSC_ADAM_EARLY=0: after the backward)
set -o pipefail
O=gpurun_out/r4sw; mkdir -p $O
for rep in 1 2 3; do
  for cfg in "1 wgrad 1" "1 start 1" "1 wgrad 0" "0 wgrad 1"; do
    set -- $cfg
    MSC_STEM_WGRAD_HALO=$1 MSC_STEM_HOOK_AT=$2 MSC_ADAM_EARLY=$3 timeout -k 10 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > $O/bench.txt 2>&1
    echo "halo=$1 hook=$2 adam_early=$3 rep=$rep rc=$? $(tail -1 $O/bench.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))')" | tee -a $O/status.txt
  done
done
