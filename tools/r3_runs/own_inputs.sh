#!/bin/bash
b() { python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms')"; }
for r in 1 2 3; do
  MSC_BENCH_OWN_INPUTS=1 b "batch in the input buffers of the trainer"
  MSC_BENCH_OWN_INPUTS=0 b "batch copied into them every step  "
done
