#!/bin/bash
# kernel table of the --fp8 bench line (which tile heights its GEMMs take)
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_f8 -o f8 -- python3 $ROOT/bench.py --fp8 --no-cpu-baseline > $ROOT/gpurun_out/r04_bench_fp8_under_rocprof.json 2> $ROOT/gpurun_out/prof_f8.err
python3 $ROOT/scripts/trace_summary.py $ROOT/gpurun_out/prof_f8/f8_kernel_trace.csv > $ROOT/gpurun_out/r04_fp8_kernel_table.txt
rm -rf $ROOT/gpurun_out/prof_f8
head -12 $ROOT/gpurun_out/r04_fp8_kernel_table.txt | cut -c1-150
