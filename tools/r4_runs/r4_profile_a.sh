#!/bin/bash
# round 4: step composition of the current build (kernel trace by stream, gaps) + what-if table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=gpurun_out/r4p; rm -rf $R; mkdir -p $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -o run -- python3 bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-parity --alt-frames-per-gpu 0 > $R/kt.log 2>&1 \
 && python3 tools/trace_step.py $R/kt/run_kernel_trace.csv 16 > $R/step_by_stream.txt && python3 tools/trace_gaps.py $R/kt/run_kernel_trace.csv > $R/step_gaps.txt
cp $R/kt/run_kernel_stats.csv $R/kernel_stats.csv; rm -f $R/kt/run_kernel_trace.csv
head -60 $R/step_by_stream.txt
cat $R/step_gaps.txt | head -40
python3 tools/whatif.py 'wgrad' 'msc_conv2d_wgrad|msc_stem_s2d_conv_wgrad' 'msc_linear_wgrad' 'msc_linear_fwd' 'msc_linear_dgrad' 'msc_conv2d_fwd|msc_stem_s2d_conv_fwd' 'msc_conv2d_dgrad' 'msc_bn_bwd' 'msc_bn_apply|msc_bn_finalize' 'msc_attention_bwd' 'msc_attention_fwd' 'msc_layernorm_fwd' 'msc_layernorm_bwd' 'msc_adam_step|msc_permute3' 'msc_pillar' 'msc_bn_relu_pool' > $R/whatif.txt 2>&1; tail -18 $R/whatif.txt
