#!/bin/bash
for r in 1 2; do
for v in 1 0; do
  MSC_BENCH_OWN_INPUTS=$v python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline --host-inputs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['host_inputs']; print('own_inputs=$v device-resident', round(d['ms_per_step'],3), 'ms | host inputs', round(h['ms_per_step'],3), 'ms ratio', round(h['vs_device_resident'],3))"
done; done
