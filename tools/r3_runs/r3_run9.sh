#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
b() { python3 bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), 'ms')"; }
for rep in 1 2; do
  for W in 8 4 5 6; do MSC_ATT_BWD_WAVES=$W b "att_bwd_waves=$W"; done
  MSC_LIB_PATH=$GRAFT_REPO_ROOT/multimodal_sc_amd/csrc/libmsc_hip_ln2.so b "ln_ur=2"
done | tee gpurun_out/r3i/ab.txt
MSC_LIB_PATH=$GRAFT_REPO_ROOT/multimodal_sc_amd/csrc/libmsc_hip_ln2.so timeout -k 10 200 python3 tools/determinism_probe.py C3 64 2>&1 | tail -4 | tee gpurun_out/r3i/ln2_determinism.txt
