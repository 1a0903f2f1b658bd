#!/bin/bash
# round 4 experiment: tile height of the two epilogue-heavy ping-pong launches (fc2's dgrad x stored derivative: MSC_PP_DREF_ROWS; fc1 forward with the
# second output: MSC_PP_C2_ROWS) inside the 64-frame step, interleaved reps on one box
set -o pipefail
R=gpurun_out/r4ppe; rm -rf $R; mkdir -p $R
run() { timeout -k 10 200 python3 bench.py --batch 64 --steps 40 --warmup 6 --no-cpu-baseline --no-roofline --no-parity --alt-frames-per-gpu 0 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
for rep in 1 2 3; do
  echo "rep $rep default $(run)"
  echo "rep $rep dref128 $(MSC_PP_DREF_ROWS=128 run)"
  echo "rep $rep c2_128 $(MSC_PP_C2_ROWS=128 run)"
  echo "rep $rep dref128+c2_128 $(MSC_PP_DREF_ROWS=128 MSC_PP_C2_ROWS=128 run)"
  echo "rep $rep dref192 $(MSC_PP_DREF_ROWS=192 run)"
done | tee $R/ab.txt
