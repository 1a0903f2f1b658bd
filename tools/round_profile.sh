#!/bin/bash
# round 4: the profiles committed under profiles/r4_* (kernel trace + stats of the normal multi-stream bench and of the single-stream
# instrumented passes, PMC passes in their own runs, step composition).  The -m gpu suite / smoke / bench line: tools/r4_runs/r4_final_a.sh;
# what-if table, host overhead, host-inputs leg: tools/r4_runs/r4_final_c.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r4r; rm -rf $R; mkdir -p $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -o run -- python3 bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-parity --alt-frames-per-gpu 0 > $R/kt.log 2>&1 \
 && python3 tools/trace_step.py $R/kt/run_kernel_trace.csv 14 > $R/step_by_stream.txt && python3 tools/trace_gaps.py $R/kt/run_kernel_trace.csv > $R/step_gaps.txt \
 && cp $R/kt/run_kernel_stats.csv $R/kernel_stats.csv && rm -f $R/kt/run_kernel_trace.csv \
 && rocprofv3 --kernel-trace --stats --output-format csv -d $R/ki -o run -- python3 bench.py --instrumented-only 8 > $R/ki.json 2> $R/ki.err \
 && cp $R/ki/run_kernel_stats.csv $R/kernel_stats_instrumented.csv && python3 tools/stats_vs_bench.py $R/kernel_stats_instrumented.csv 9 $R/ki.json | tee $R/stats_vs_bench.txt \
 && rm -f $R/ki/run_kernel_trace.csv || exit 1
export PMC_STEPS=5
B="python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline --no-parity --alt-frames-per-gpu 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pf -o run -- $B > $R/pf.log 2>&1 \
 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pw -o run -- $B > $R/pw.log 2>&1 \
 && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/ps -o run -- $B > $R/ps.log 2>&1 \
 || exit 2
python3 tools/pmc_hbm_summary.py $R/pf/run_counter_collection.csv $R/pw/run_counter_collection.csv $R/pmc_hbm.csv
python3 tools/pmc_summary.py $R/ps/run_counter_collection.csv > $R/pmc_sq.csv
rm -f $R/p?/run_counter_collection.csv
cat $R/pmc_hbm.csv.meta.json; echo; head -12 $R/pmc_sq.csv | cut -c1-250
