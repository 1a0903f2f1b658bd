#!/bin/bash
set -o pipefail
O=gpurun_out/r4c; mkdir -p $O
for rep in 1 2; do
  for on in 1 0; do
    MSC_CONV_C64=$on timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > $O/bench_${on}_$rep.txt 2>&1
    echo "c64=$on rep=$rep rc=$? $(tail -1 $O/bench_${on}_$rep.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))')" | tee -a $O/status.txt
  done
done
