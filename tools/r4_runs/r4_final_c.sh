#!/bin/bash
# round 4, end-of-round call 3: what-if table (step time with one kernel family skipped), host enqueue time, host-inputs leg of the bench
set -o pipefail
R=gpurun_out/r4fc; rm -rf $R; mkdir -p $R
timeout -k 10 700 python3 tools/whatif.py 'wgrad' 'msc_conv2d_wgrad|msc_stem_s2d_conv_wgrad' 'msc_linear_wgrad' 'msc_linear_fwd' 'msc_linear_dgrad' 'msc_conv2d_fwd|msc_stem_s2d_conv_fwd' 'msc_conv2d_dgrad' 'msc_bn_bwd' 'msc_bn_apply|msc_bn_finalize' 'msc_attention_bwd' 'msc_attention_fwd' 'msc_layernorm_bwd' 'msc_adam_step|msc_permute3' 'msc_pillar' 'msc_bn_relu_pool' > $R/whatif.txt 2>&1; echo "whatif rc=$?"; tail -17 $R/whatif.txt
timeout -k 10 200 python3 tools/host_overhead.py 64 2>&1 | tail -1 | tee $R/host_overhead.txt; MSC_REPLAY=0 timeout -k 10 200 python3 tools/host_overhead.py 64 2>&1 | tail -1 | tee -a $R/host_overhead.txt
timeout -k 10 400 python3 bench.py --host-inputs --no-cpu-baseline > $R/bench_host_inputs.json 2>/dev/null; echo "bench rc=$?"; python3 -c "import json; j=json.loads(open('$R/bench_host_inputs.json').read().strip().splitlines()[-1]); print({k: j[k] for k in j if 'host' in k or k in ('value','ms_per_step')})"
