for i in 1 2; do
OCC_LIB=$PWD/occm_amd/libocc_hip_base.so python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
OCC_LIB=$PWD/occm_amd/libocc_hip_base.so python bench.py --xlsr 1b --backend senet --bs 32 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base cfg4', d['value'], d['ms_per_step'])"
python bench.py --xlsr 1b --backend senet --bs 32 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new  cfg4', d['value'], d['ms_per_step'])"
