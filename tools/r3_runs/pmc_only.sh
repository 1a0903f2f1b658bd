#!/bin/bash
# the two HBM PMC passes (separate runs, as the guide prescribes) + summary, for the sidecar of the final sources
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/pmc; rm -rf $R; mkdir -p $R
export PMC_STEPS=5
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pf -o run -- python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline > $R/pf.log 2>&1 \
 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pw -o run -- python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline > $R/pw.log 2>&1
python3 tools/pmc_hbm_summary.py $R/pf/run_counter_collection.csv $R/pw/run_counter_collection.csv $R/pmc_hbm.csv
rm -f $R/p?/run_counter_collection.csv
cat $R/pmc_hbm.csv.meta.json; echo; head -4 $R/pmc_hbm.csv
