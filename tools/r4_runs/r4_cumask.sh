#!/bin/bash
set -o pipefail
O=gpurun_out/r4cm; mkdir -p $O
L=$PWD/multimodal_sc_amd/csrc/libmsc_hip_cumask.so
for rep in 1 2; do
  for m in none 55555555 77777777 0f0f0f0f ffff0000; do
    if [ $m = none ]; then unset MSC_SIDE_CU_MASK; else export MSC_SIDE_CU_MASK=$m; fi
    MSC_C64W_WGS=128 MSC_LIB_PATH=$L timeout -k 10 400 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > $O/bench.txt 2>&1
    echo "side_cu_mask=$m rep=$rep rc=$? $(tail -1 $O/bench.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))')" | tee -a $O/status.txt
  done
done
