#!/bin/bash
# round 4: weight-gradient split target 512 against 640 workgroups, 64 and 32 frames per GPU, interleaved reps on one box
set -o pipefail
R=gpurun_out/r4w512; rm -rf $R; mkdir -p $R
run() { timeout -k 10 200 python3 bench.py --batch $1 --steps 40 --warmup 6 --no-cpu-baseline --no-roofline --no-parity --alt-frames-per-gpu 0 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
for rep in 1 2 3; do
  for W in 640 512 576 448; do echo "rep $rep frames 64 MSC_WGRAD_BLOCKS=$W $(MSC_WGRAD_BLOCKS=$W run 64)"; done
  for W in 640 512; do echo "rep $rep frames 32 MSC_WGRAD_BLOCKS=$W $(MSC_WGRAD_BLOCKS=$W run 32)"; done
done | tee $R/ab.txt
