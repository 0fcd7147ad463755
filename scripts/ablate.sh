export TMPDIR=/tmp
for v in 0 1 2 4 3 5 6 7; do echo "== ZV_DBG=$v"; ZV_DBG=$v python bench.py --steps 50 --no-cpu-baseline --no-graph 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], [(k['name'], round(k['avg_us'],1)) for k in j['extra']['kernels']])"; done
