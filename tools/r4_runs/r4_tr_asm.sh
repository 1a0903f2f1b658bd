#!/bin/bash
# round 4: transposed LDS reads of the mn-major operands (weight gradients) issued as inline assembly, so that the compiler no longer places
# s_waitcnt vmcnt(0) between the next stage's DMA issue and this stage's fragment reads.  Parity + bitwise tests, the contraction microbenchmark,
# and the step with the old library (.ab_old/libmsc_hip.so via MSC_LIB_PATH) against the new one, interleaved on one box.
set -o pipefail
R=gpurun_out/r4tr; mkdir -p $R
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemm_shapes_gpu.py -m gpu -q -x --deselect tests/test_kernels_gpu.py::test_loaded_library_was_built_from_checked_out_sources > $R/tests1.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $R/tests1.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/gemm_bench.py > $R/gemm_bench_after.txt 2>&1 || exit 3
grep -i "wgrad" $R/gemm_bench_after.txt | cut -c1-120
run() { timeout -k 10 200 python3 bench.py --batch $1 --steps 40 --warmup 6 --no-cpu-baseline --no-roofline --no-parity --alt-frames-per-gpu 0 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
for rep in 1 2 3; do
  echo "rep $rep frames 64 old $(MSC_LIB_PATH=$PWD/.ab_old/libmsc_hip.so run 64)   new $(run 64)"
  echo "rep $rep frames 32 old $(MSC_LIB_PATH=$PWD/.ab_old/libmsc_hip.so run 32)   new $(run 32)"
done | tee $R/step_ab.txt
