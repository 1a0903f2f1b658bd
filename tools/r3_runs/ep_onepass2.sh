#!/bin/bash
b() { python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms')"; }
export MSC_GELU_DSAVE=0
for r in 1 2 3; do
  b "one pass + hoisted loads"
  MSC_LIB_PATH=multimodal_sc_amd/csrc/libmsc_hip_ep2.so b "64-row passes + hoisted loads"
  MSC_LIB_PATH=multimodal_sc_amd/csrc/libmsc_hip_oldgemm.so b "previous epilogue"
done
