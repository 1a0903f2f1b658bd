#!/bin/bash
# round 4: 128-row tile of the ping-pong kernel (gemm_pp_kernel<2,2>): parity + bitwise tests of the shape file's linear cases, the dense linears at
# the 32-frame shard (9792 rows) and at 64 frames, then the bench line's alt geometry with the tile allowed / not allowed (knob via env: MSC_GEMM_PP)
set -o pipefail
R=gpurun_out/r4pp128; rm -rf $R; mkdir -p $R
timeout -k 10 500 python -m pytest tests/test_gemm_shapes_gpu.py -m gpu -q -x -k "fusion_linear or ping_pong" > $R/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $R/tests.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 tools/pp_check.py 9792 5 > $R/pp_check_9792.txt 2>&1 || exit 3; grep -v amdgpu.ids $R/pp_check_9792.txt | cut -c1-330
timeout -k 10 200 python3 tools/pp_check.py 19584 3 > $R/pp_check_19584.txt 2>&1 || exit 4; grep TOTAL $R/pp_check_19584.txt
for rep in 1 2; do
  timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-parity > $R/b_$rep.json 2>/dev/null || exit 5
  python3 -c "import json; j=json.loads(open('$R/b_$rep.json').read().strip().splitlines()[-1]); print('rep $rep  64: %.3f ms   32: %.3f ms' % (j['ms_per_step'], j['alt_geometry']['ms_per_step']))"
done | tee $R/ab.txt
