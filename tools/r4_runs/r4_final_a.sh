#!/bin/bash
# round 4, end-of-round: the -m gpu suite in two halves (PART=1: kernel + shape tests, PART=2: model / trainer / DP / frames + smoke() + the default bench line)
set -o pipefail
R=gpurun_out/r4fa; mkdir -p $R
if [ "${PART:-1}" = 1 ]; then
  timeout -k 10 1100 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_shapes_gpu.py -m gpu -q -x > $R/tests1.log 2>&1; rc=$?; echo "tests part 1 rc=$rc"; tail -3 $R/tests1.log | cut -c1-300
  exit $rc
fi
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --deselect tests/test_kernels_gpu.py --deselect tests/test_gemm_shapes_gpu.py > $R/tests2.log 2>&1; rc=$?; echo "tests part 2 rc=$rc"; tail -3 $R/tests2.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $R/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $R/smoke.log
timeout -k 10 600 python bench.py > $R/bench.json 2> $R/bench.err; echo "bench rc=$?"; tail -c 1500 $R/bench.json
