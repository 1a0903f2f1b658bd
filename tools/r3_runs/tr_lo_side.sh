#!/bin/bash
python3 -m pytest tests/test_model_gpu.py -q -m gpu -x -k "determinism or replay or dp or graph or checkpoint" 2>&1 | tail -2
b() { python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms')"; }
for r in 1 2 3; do
  MSC_TR_LO_SIDE=1 b "side=1"
  MSC_TR_LO_SIDE=0 b "side=0"
done
