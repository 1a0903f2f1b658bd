#!/bin/bash
# step A/B of alternative library builds: r4_ab.sh <tag> <alt lib suffix> [reps]
set -o pipefail
O=gpurun_out/$1; mkdir -p $O
for rep in $(seq 1 ${3:-2}); do
  for lib in base $2; do
    L=$PWD/multimodal_sc_amd/csrc/libmsc_hip.so; [ $lib = base ] || L=$PWD/multimodal_sc_amd/csrc/libmsc_hip_$lib.so
    MSC_LIB_PATH=$L timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > $O/bench_${lib}_$rep.txt 2>&1
    echo "$lib rep=$rep rc=$? $(tail -1 $O/bench_${lib}_$rep.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))')" | tee -a $O/status.txt
  done
done
