#!/bin/bash
# round 4, after the weight gradients stopped waiting for their own prefetch: side-stream knobs on the new balance (information for the next round; defaults unchanged)
set -o pipefail
R=gpurun_out/r4ek; rm -rf $R; mkdir -p $R
run() { timeout -k 10 200 python3 bench.py --batch 64 --steps 40 --warmup 6 --no-cpu-baseline --no-roofline --no-parity --alt-frames-per-gpu 0 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
for rep in 1 2; do
  echo "rep $rep default $(run)"
  for W in 384 640 768; do echo "rep $rep MSC_WGRAD_BLOCKS=$W $(MSC_WGRAD_BLOCKS=$W run)"; done
  for W in 96 192 256; do echo "rep $rep MSC_C64W_WGS=$W $(MSC_C64W_WGS=$W run)"; done
  echo "rep $rep MSC_SIDE_PRIO=0 $(MSC_SIDE_PRIO=0 run)"
done | tee $R/knobs.txt
