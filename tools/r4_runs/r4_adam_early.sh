#!/bin/bash
# encoder half of the optimizer update on the side stream beside the stem's weight gradient (MSC_ADAM_EARLY=1) vs on the main stream at the end
set -o pipefail
O=gpurun_out/r4ae; mkdir -p $O
for rep in 1 2 3 4; do
  for E in 1 0; do
    MSC_ADAM_EARLY=$E timeout -k 10 400 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > $O/bench_${E}_$rep.txt 2>&1
    echo "adam_early=$E rep=$rep rc=$? $(tail -1 $O/bench_${E}_$rep.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))')" | tee -a $O/status.txt
  done
done
