#!/bin/bash
# kernel-trace of the bench with and without the four-wave GEMM (GEMM rows only)
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for q in 0 1; do
  export OCC_GEMM_Q4=$q
  rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_q4_$q -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2> $ROOT/gpurun_out/prof_q4_$q.err
  echo "== OCC_GEMM_Q4=$q"
  python3 $ROOT/scripts/trace_summary.py $ROOT/gpurun_out/prof_q4_$q/bench_kernel_trace.csv "gemm_p8|gemm_q4" 12
  rm -rf $ROOT/gpurun_out/prof_q4_$q
done
