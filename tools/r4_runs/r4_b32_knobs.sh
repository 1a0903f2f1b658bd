#!/bin/bash
# round 4: the weight-gradient split target at 32 frames per GPU (the side stream has slack there: 2.5 ms busy of a 4.85 ms step) — interleaved reps
set -o pipefail
R=gpurun_out/r4b32k; rm -rf $R; mkdir -p $R
run() { timeout -k 10 200 python3 bench.py --batch 32 --steps 40 --warmup 6 --no-cpu-baseline --no-roofline --no-parity --alt-frames-per-gpu 0 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
for rep in 1 2; do
  for W in 640 192 256 384 512; do echo "rep $rep MSC_WGRAD_BLOCKS=$W $(MSC_WGRAD_BLOCKS=$W run)"; done
  echo "rep $rep MSC_SIDE_PRIO=0 $(MSC_SIDE_PRIO=0 run)"
done | tee $R/knobs.txt
