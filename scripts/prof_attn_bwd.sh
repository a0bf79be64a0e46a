#!/bin/bash
# per-kernel durations of scripts/time_attn_bwd.py (the split attention backward's two kernels against the one-kernel form)
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_ab -o ab -- python3 $ROOT/scripts/time_attn_bwd.py > $ROOT/gpurun_out/prof_ab.log 2>&1
python3 $ROOT/scripts/trace_summary.py $ROOT/gpurun_out/prof_ab/ab_kernel_trace.csv "attention" 12
rm -rf $ROOT/gpurun_out/prof_ab
