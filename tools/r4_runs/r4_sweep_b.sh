#!/bin/bash
mkdir -p gpurun_out/r4s
run() { env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3))'; }
for rep in 1 2; do
  for pad in 0 16384 32768 65536 98304; do echo "rep $rep MSC_WGRAD_LDS_PAD=$pad $(run MSC_WGRAD_LDS_PAD=$pad)" | tee -a gpurun_out/r4s/sweep_b.txt; done
done
