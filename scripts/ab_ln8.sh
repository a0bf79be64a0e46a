cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in 1024 100000 1024 100000; do OCC_LN8_BLOCKS=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp8 ln8_blocks=$v', j['value'], j['ms_per_step'])"; done
for v in 100000 1024; do OCC_LN8_BLOCKS=$v python bench.py --no-cpu-baseline --xlsr 1b --backend senet --bs 32 --fp8 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 fp8 ln8_blocks=$v', j['value'], j['ms_per_step'])"; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', j['value'], j['ms_per_step'])"
python bench.py --no-cpu-baseline --xlsr 1b --backend senet --bs 32 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 bf16', j['value'], j['ms_per_step'])"
