#!/bin/bash
# kernel-trace + stats of the normal bench and of the instrumented passes (no PMC, no tests): the final build's numbers for profiles/r3_kernel_stats*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r3s; rm -rf $R; mkdir -p $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -o run -- python3 bench.py --steps 12 --warmup 5 --no-cpu-baseline > $R/kt.log 2>&1 \
 && python3 tools/trace_step.py $R/kt/run_kernel_trace.csv 14 > $R/step_by_stream.txt && python3 tools/trace_gaps.py $R/kt/run_kernel_trace.csv > $R/step_gaps.txt
cp $R/kt/run_kernel_stats.csv $R/kernel_stats.csv; rm -f $R/kt/run_kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/ki -o run -- python3 bench.py --instrumented-only 8 > $R/ki.json 2> $R/ki.err \
 && cp $R/ki/run_kernel_stats.csv $R/kernel_stats_instrumented.csv && python3 tools/stats_vs_bench.py $R/kernel_stats_instrumented.csv 9 $R/ki.json | tee $R/stats_vs_bench.txt
rm -f $R/ki/run_kernel_trace.csv
python3 tools/whatif.py 'wgrad' 'msc_linear_fwd' 'msc_linear_dgrad' 'msc_conv2d_fwd|msc_stem_s2d_conv_fwd' 'msc_conv2d_dgrad' 'msc_bn_bwd' 'msc_bn_apply|msc_bn_finalize' 'msc_attention_bwd' 'msc_attention_fwd' 'msc_layernorm_bwd' 'msc_adam_step|msc_permute3' 'msc_pillar' 'msc_bn_relu_pool' > $R/whatif.txt 2>&1; tail -14 $R/whatif.txt
head -3 $R/step_by_stream.txt; head -8 $R/step_gaps.txt
