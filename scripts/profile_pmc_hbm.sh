#!/bin/bash
# HBM-traffic counter passes (FETCH_SIZE and WRITE_SIZE in separate runs, no tracing domains) of one bench.py configuration:
#   gpurun --timeout 900 -- 'bash scripts/profile_pmc_hbm.sh r03 frozen --frozen'
# -> gpurun_out/<tag>_pmc_hbm_<name>.json (copy into profiles/ so that bench.py's roofline.traffic finds it) and the bench line of that
# configuration in gpurun_out/<tag>_bench_<name>.json.
set -e
TAG=$1; NAME=$2; shift 2
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/prof_fetch -o bench -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2> $ROOT/gpurun_out/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/prof_write -o bench -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2> $ROOT/gpurun_out/prof_write.err
cd $ROOT
python3 scripts/pmc_summary.py gpurun_out/prof_fetch/bench_counter_collection.csv gpurun_out/prof_write/bench_counter_collection.csv gpurun_out/${TAG}_pmc_hbm_${NAME}.json "$(python3 bench.py --print-workload "$@")"
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write
cp gpurun_out/${TAG}_pmc_hbm_${NAME}.json profiles/
python3 bench.py --no-cpu-baseline "$@" > gpurun_out/${TAG}_bench_${NAME}.json 2> gpurun_out/${TAG}_bench_${NAME}.err
cat gpurun_out/${TAG}_bench_${NAME}.json
