#!/bin/bash
# split-DMA build (default lib) vs the whole-piece-in-the-read-part build (libmsc_hip_nosplit.so): tests, per-shape times, step A/B
set -o pipefail
O=gpurun_out/r4i
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gemm_shapes_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "linear or ping_pong or gelu or determinism" > $O/tests.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt
tail -4 $O/tests.txt | cut -c1-250
timeout -k 10 600 python tools/pp_check.py 19584 3 > $O/pp_check_split.txt 2>&1; grep -E "fwd|dgrd|TOTAL|8192|4096x" $O/pp_check_split.txt | cut -c1-330
MSC_LIB_PATH=$PWD/multimodal_sc_amd/csrc/libmsc_hip_nosplit.so timeout -k 10 600 python tools/pp_check.py 19584 3 > $O/pp_check_nosplit.txt 2>&1; grep -E "fwd|dgrd|TOTAL|8192|4096x" $O/pp_check_nosplit.txt | cut -c1-330
for rep in 1 2; do
  for lib in split nosplit; do
    L=$PWD/multimodal_sc_amd/csrc/libmsc_hip.so; [ $lib = nosplit ] && L=$PWD/multimodal_sc_amd/csrc/libmsc_hip_nosplit.so
    MSC_LIB_PATH=$L timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > $O/bench_${lib}_$rep.txt 2>&1
    echo "$lib rep=$rep rc=$? $(tail -1 $O/bench_${lib}_$rep.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))')" | tee -a $O/status.txt
  done
done
