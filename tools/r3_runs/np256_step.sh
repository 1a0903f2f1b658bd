#!/bin/bash
# the 256x128 one-tile-per-workgroup variant (experiment build) inside the step, where the fusion GEMMs are half-batch (9792 rows: 39 x 12 tiles = one round)
b() { python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms')"; }
export MSC_LIB_PATH=multimodal_sc_amd/csrc/libmsc_hip_np256.so
for r in 1 2 3; do
  MSC_GEMM_NP256=0 b "128x128"
  MSC_GEMM_NP256=1 b "256x128 where eligible"
done
