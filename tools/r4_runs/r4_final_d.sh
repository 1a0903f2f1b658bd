#!/bin/bash
# round 4, last call: the profiles of the final build (tools/round_profile.sh), its PMC summary put where bench.py reads it, then the second half of the
# -m gpu suite, smoke() and the default bench line (tools/r4_runs/r4_final_a.sh PART=2)
set -o pipefail
bash tools/round_profile.sh > gpurun_out/round_profile.log 2>&1 || { echo "round_profile failed"; tail -5 gpurun_out/round_profile.log; exit 1; }
tail -4 gpurun_out/round_profile.log | cut -c1-250
cp gpurun_out/r4r/pmc_hbm.csv profiles/r4_pmc_hbm_bench_c3.csv && cp gpurun_out/r4r/pmc_hbm.csv.meta.json profiles/r4_pmc_hbm_bench_c3.csv.meta.json
PART=2 bash tools/r4_runs/r4_final_a.sh
