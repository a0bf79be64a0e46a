#!/bin/bash
# End-of-round validation on one GPU box: the whole GPU suite, smoke(), the headline bench line and the other configurations' lines.
#   gpurun --timeout 1190 -- 'bash scripts/final_round.sh r04'
TAG=${1:-r04}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/${TAG}_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err && cut -c1-160 gpurun_out/${TAG}_bench.json
python bench.py --no-cpu-baseline --frozen > gpurun_out/${TAG}_bench_frozen.json 2>/dev/null && cut -c1-160 gpurun_out/${TAG}_bench_frozen.json
python bench.py --no-cpu-baseline --fp8 > gpurun_out/${TAG}_bench_fp8.json 2>/dev/null && cut -c1-160 gpurun_out/${TAG}_bench_fp8.json
python bench.py --no-cpu-baseline --ssl-dropouts 0.1,0.1,0.1,0.05 > gpurun_out/${TAG}_bench_ssl_dropouts.json 2>/dev/null && cut -c1-160 gpurun_out/${TAG}_bench_ssl_dropouts.json
python bench.py --no-cpu-baseline --xlsr 1b --backend senet --bs 32 > gpurun_out/${TAG}_bench_cfg4_bf16.json 2>/dev/null && cut -c1-160 gpurun_out/${TAG}_bench_cfg4_bf16.json
python bench.py --no-cpu-baseline --xlsr 1b --backend senet --bs 32 --fp8 > gpurun_out/${TAG}_bench_cfg4_fp8.json 2>/dev/null && cut -c1-160 gpurun_out/${TAG}_bench_cfg4_fp8.json
python bench.py --no-cpu-baseline --frozen --backend senet > gpurun_out/${TAG}_bench_frozen_senet.json 2>/dev/null && cut -c1-160 gpurun_out/${TAG}_bench_frozen_senet.json
python scripts/bench_score.py --n 96 > gpurun_out/${TAG}_score_bench.jsonl 2> gpurun_out/${TAG}_score_bench.err; grep -c utt_per_s gpurun_out/${TAG}_score_bench.jsonl
