#!/bin/bash
# the bench lines of every configuration (final_round.sh without the test suite):  bash scripts/bench_lines.sh r04
TAG=${1:-r04}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err && cut -c1-160 gpurun_out/${TAG}_bench.json
python bench.py --no-cpu-baseline --frozen > gpurun_out/${TAG}_bench_frozen.json 2>/dev/null && cut -c1-160 gpurun_out/${TAG}_bench_frozen.json
python bench.py --no-cpu-baseline --fp8 > gpurun_out/${TAG}_bench_fp8.json 2>/dev/null && cut -c1-160 gpurun_out/${TAG}_bench_fp8.json
python bench.py --no-cpu-baseline --ssl-dropouts 0.1,0.1,0.1,0.05 > gpurun_out/${TAG}_bench_ssl_dropouts.json 2>/dev/null && cut -c1-160 gpurun_out/${TAG}_bench_ssl_dropouts.json
python bench.py --no-cpu-baseline --xlsr 1b --backend senet --bs 32 > gpurun_out/${TAG}_bench_cfg4_bf16.json 2>/dev/null && cut -c1-160 gpurun_out/${TAG}_bench_cfg4_bf16.json
python bench.py --no-cpu-baseline --xlsr 1b --backend senet --bs 32 --fp8 > gpurun_out/${TAG}_bench_cfg4_fp8.json 2>/dev/null && cut -c1-160 gpurun_out/${TAG}_bench_cfg4_fp8.json
python bench.py --no-cpu-baseline --frozen --backend senet > gpurun_out/${TAG}_bench_frozen_senet.json 2>/dev/null && cut -c1-160 gpurun_out/${TAG}_bench_frozen_senet.json
