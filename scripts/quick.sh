# usage: bash scripts/quick.sh [env assignments...]  -> one-line bench summary
export TMPDIR=/tmp
env "$@" python bench.py --steps 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print('xRT', j['value'], 'ms', j['ms_per_step'], 'frac', r['frac'], 'avg_us', r['avg_launch_us'], 'n', r['launches_per_step'], 'TF', r['mfma_TFLOPs']); print([(k['name'], k['launches_per_step'], round(k['avg_us'],1)) for k in j['extra']['kernels']])"
