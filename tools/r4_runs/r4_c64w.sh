#!/bin/bash
# layer-1 weight gradient on conv_c64w.hip: per-launch time and step A/B (MSC_CONV_C64=0: everything of layer 1 back on the implicit GEMM; here only the
# weight gradient is switched through the workgroup-count knob / a second library would be needed, so the A/B is c64w_wgs = 256 | 128 | 224 vs the
# gathered GEMM measured by tools/gemm_bench.py)
set -o pipefail
O=gpurun_out/r4cw; mkdir -p $O
for w in 64 96 128 160; do MSC_C64W_WGS=$w timeout -k 10 300 python tools/gemm_bench.py 2>&1 | grep "wgrad l1 3x3" | sed "s/^/wgs=$w /"; done
for rep in 1 2 3; do
  for w in -1 64 96 128 160; do
    MSC_C64W_WGS=$w timeout -k 10 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > $O/bench.txt 2>&1
    echo "c64w_wgs=$w rep=$rep rc=$? $(tail -1 $O/bench.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))')" | tee -a $O/status.txt
  done
done
