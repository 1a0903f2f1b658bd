#!/bin/bash
# round 3: the whole -m gpu suite, then the profiles committed under profiles/r3_* (kernel trace + stats of the normal multi-stream
# bench and of the single-stream instrumented passes, PMC passes in their own runs, step composition, what-if table)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r3r; rm -rf $R; mkdir -p $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $R/tests.log 2>&1; echo "tests rc=$?"; tail -3 $R/tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -o run -- python3 bench.py --steps 12 --warmup 5 --no-cpu-baseline > $R/kt.log 2>&1 \
 && python3 tools/trace_step.py $R/kt/run_kernel_trace.csv 14 > $R/step_by_stream.txt && python3 tools/trace_gaps.py $R/kt/run_kernel_trace.csv > $R/step_gaps.txt
cp $R/kt/run_kernel_stats.csv $R/kernel_stats.csv; rm -f $R/kt/run_kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/ki -o run -- python3 bench.py --instrumented-only 8 > $R/ki.json 2> $R/ki.err \
 && cp $R/ki/run_kernel_stats.csv $R/kernel_stats_instrumented.csv && python3 tools/stats_vs_bench.py $R/kernel_stats_instrumented.csv 9 $R/ki.json | tee $R/stats_vs_bench.txt
rm -f $R/ki/run_kernel_trace.csv
export PMC_STEPS=5
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/pf -o run -- python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline > $R/pf.log 2>&1 \
 && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/pw -o run -- python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline > $R/pw.log 2>&1 \
 && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/ps -o run -- python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline > $R/ps.log 2>&1
python3 tools/pmc_hbm_summary.py $R/pf/run_counter_collection.csv $R/pw/run_counter_collection.csv $R/pmc_hbm.csv
python3 tools/pmc_summary.py $R/ps/run_counter_collection.csv > $R/pmc_sq.csv
rm -f $R/p?/run_counter_collection.csv
cat $R/pmc_hbm.csv.meta.json; echo
python3 tools/whatif.py 'wgrad' 'msc_conv2d_wgrad|msc_stem_s2d_conv_wgrad' 'msc_linear_wgrad' 'msc_linear_fwd' 'msc_linear_dgrad' 'msc_conv2d_fwd|msc_stem_s2d_conv_fwd' 'msc_conv2d_dgrad' 'msc_bn_bwd' 'msc_bn_apply|msc_bn_finalize' 'msc_attention_bwd' 'msc_attention_fwd' 'msc_layernorm_bwd' 'msc_adam_step|msc_permute3' 'msc_pillar' 'msc_bn_relu_pool' > $R/whatif.txt 2>&1; tail -16 $R/whatif.txt
python3 tools/host_overhead.py 64 2>&1 | tail -1 | tee $R/host_overhead.txt; MSC_REPLAY=0 python3 tools/host_overhead.py 64 2>&1 | tail -1 | tee -a $R/host_overhead.txt
python3 tools/host_overhead.py 16 2>&1 | tail -1 | tee -a $R/host_overhead.txt; MSC_REPLAY=0 python3 tools/host_overhead.py 16 2>&1 | tail -1 | tee -a $R/host_overhead.txt
python3 bench.py --host-inputs > $R/bench.json 2>/dev/null; tail -c 900 $R/bench.json
