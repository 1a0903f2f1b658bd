#!/bin/bash
# round 4: which of the two halo weight-gradient kernels (stem, layer 1) should take the inline-assembly transposed reads — four libraries, interleaved, one box
set -o pipefail
R=gpurun_out/r4trh; rm -rf $R; mkdir -p $R
run() { timeout -k 10 200 python3 bench.py --batch 64 --steps 40 --warmup 6 --no-cpu-baseline --no-roofline --no-parity --alt-frames-per-gpu 0 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))"; }
for rep in 1 2 3; do
  echo "rep $rep gemm_only $(MSC_LIB_PATH=$PWD/.ab_old/lib_gemm_only.so run)  +stem $(MSC_LIB_PATH=$PWD/.ab_old/lib_stem.so run)  +c64w $(MSC_LIB_PATH=$PWD/.ab_old/lib_c64w.so run)  +both $(run)"
done | tee $R/ab.txt
