#!/bin/bash
# Round profile of the headline command (bench.py, BASELINE configs[2]); run on the GPU box from the repo root, e.g.
#   gpurun --timeout 1100 -- 'bash scripts/profile_round.sh r02'
# 1) kernel-trace + stats of the bench command; 2)-4) separate PMC passes (HBM read bytes, HBM write bytes, MFMA utilisation), never
# combined with tracing domains.  Summaries land in gpurun_out/<tag>_*; copy the ones to keep into profiles/.
set -e
TAG=${1:-r02}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_stats -o bench -- python3 $ROOT/bench.py --no-cpu-baseline > $ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2> $ROOT/gpurun_out/prof_stats.err
cp $ROOT/gpurun_out/prof_stats/bench_kernel_stats.csv $ROOT/gpurun_out/${TAG}_bench_kernel_stats.csv
python3 $ROOT/scripts/trace_summary.py $ROOT/gpurun_out/prof_stats/bench_kernel_trace.csv > $ROOT/gpurun_out/${TAG}_bench_kernel_table.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/prof_fetch -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/prof_write -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/prof_write.err
rocprofv3 --pmc MfmaUtil GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/prof_mfma -o bench -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $ROOT/gpurun_out/prof_mfma.err
cd $ROOT
python3 scripts/pmc_mfma_summary.py gpurun_out/prof_mfma/bench_counter_collection.csv gpurun_out/${TAG}_pmc_mfma.json
python3 scripts/pmc_summary.py gpurun_out/prof_fetch/bench_counter_collection.csv gpurun_out/prof_write/bench_counter_collection.csv gpurun_out/${TAG}_pmc_hbm_finetune.json "$(python3 bench.py --print-workload)"
rm -rf gpurun_out/prof_mfma gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_stats
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cat gpurun_out/${TAG}_bench.json
