#!/bin/bash
# the three PMC passes of tools/round_profile.sh on their own (each counter set in its own run, no other tracing)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r4r; mkdir -p $R; rm -rf $R/pf $R/pw $R/ps
export PMC_STEPS=5
B="python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline --no-parity --alt-frames-per-gpu 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pf -o run -- $B > $R/pf.log 2>&1 \
 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pw -o run -- $B > $R/pw.log 2>&1 \
 && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/ps -o run -- $B > $R/ps.log 2>&1 \
 || exit 2
python3 tools/pmc_hbm_summary.py $R/pf/run_counter_collection.csv $R/pw/run_counter_collection.csv $R/pmc_hbm.csv
python3 tools/pmc_summary.py $R/ps/run_counter_collection.csv > $R/pmc_sq.csv
rm -f $R/p?/run_counter_collection.csv
cat $R/pmc_hbm.csv.meta.json; echo; head -8 $R/pmc_sq.csv | cut -c1-250; head -8 $R/pmc_hbm.csv | cut -c1-250
