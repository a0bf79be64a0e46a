for i in 1 2 3; do
OCC_GELU_KEEP_GRAD=0 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old pair', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keep grad', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
