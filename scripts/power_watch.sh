# diagnostic: board power and shader clock while the bench runs (is the step power-bound?)
# usage (GPU box): bash scripts/power_watch.sh
python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-extras > /tmp/pw_bench.json 2>/dev/null &
BP=$!
sleep 22
for i in $(seq 12); do
  rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk|Max Graphics" | tr '\n' ' '; echo
  sleep 0.5
done
wait $BP
python -c "
import json
d=json.loads(open('/tmp/pw_bench.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
