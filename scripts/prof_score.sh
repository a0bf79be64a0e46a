#!/bin/bash
# kernel table of the f32 scoring path in masked batches of 16
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_sc -o sc -- python3 $ROOT/scripts/bench_score.py --n 64 --dtypes f32 --batches 16 > $ROOT/gpurun_out/prof_sc.log 2>&1
grep utt_per_s $ROOT/gpurun_out/prof_sc.log | cut -c1-200
python3 $ROOT/scripts/trace_summary.py $ROOT/gpurun_out/prof_sc/sc_kernel_trace.csv "" 30
rm -rf $ROOT/gpurun_out/prof_sc
