#!/bin/bash
# pref-DMA build (default lib) vs the whole-piece-in-the-read-part build (libmsc_hip_nopref.so): tests, per-shape times, step A/B
set -o pipefail
O=gpurun_out/r4k
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gemm_shapes_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "linear or ping_pong or gelu or determinism" > $O/tests.txt 2>&1; echo "tests rc=$?" | tee -a $O/status.txt
tail -4 $O/tests.txt | cut -c1-250
timeout -k 10 600 python tools/pp_check.py 19584 3 > $O/pp_check_pref.txt 2>&1; grep -E "fwd|dgrd|TOTAL|8192|4096x" $O/pp_check_pref.txt | cut -c1-330
MSC_LIB_PATH=$PWD/multimodal_sc_amd/csrc/libmsc_hip_nopref.so timeout -k 10 600 python tools/pp_check.py 19584 3 > $O/pp_check_nopref.txt 2>&1; grep -E "fwd|dgrd|TOTAL|8192|4096x" $O/pp_check_nopref.txt | cut -c1-330
for rep in 1 2; do
  for lib in pref nopref; do
    L=$PWD/multimodal_sc_amd/csrc/libmsc_hip.so; [ $lib = nopref ] && L=$PWD/multimodal_sc_amd/csrc/libmsc_hip_nopref.so
    MSC_LIB_PATH=$L timeout -k 10 400 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --alt-frames-per-gpu 0 --no-parity > $O/bench_${lib}_$rep.txt 2>&1
    echo "$lib rep=$rep rc=$? $(tail -1 $O/bench_${lib}_$rep.txt | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(round(j["ms_per_step"],3), round(j["psnr_db"],4), round(j["loss"],5))')" | tee -a $O/status.txt
  done
done
for lib in pref nopref; do
  L=$PWD/multimodal_sc_amd/csrc/libmsc_hip.so; [ $lib = nopref ] && L=$PWD/multimodal_sc_amd/csrc/libmsc_hip_nopref.so
  MSC_LIB_PATH=$L timeout -k 10 300 python bench.py --instrumented-only 6 > $O/instr_$lib.json 2>/dev/null
  echo "$lib instrumented: $(python -c 'import sys,json; j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=j.get("breakdown_ms_per_step") or j.get("breakdown") or {}; print({k: b[k] for k in ("msc_linear_fwd","msc_linear_dgrad") if k in b} or list(j)[:12])' $O/instr_$lib.json)" | tee -a $O/status.txt
done
