# serving-loop A/B on ONE box (interleaved rounds): batches in flight x concurrency limit x tail groups
# usage (GPU box): bash scripts/ab_serving.sh [rounds]   -> one line per arm and round: ms_per_step, gpu idle per step
R=${1:-3}
run() {   # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
gi=d['extra'].get('gpu_idle') or {}
print('%-34s ms_per_step %.3f  xRT %.0f  gpu_idle_ms_per_step %s' % ('$name', d['ms_per_step'], d['value'], gi.get('idle_ms_per_step')))"
}
for r in $(seq $R); do
  run "lanes2"                      ZV_BENCH_LANES=2
  run "lanes3_conc2 (default)"      ZV_BENCH_LANES=3
  run "lanes3_conc2_tail1"          ZV_BENCH_LANES=3 ZV_TAIL_GROUPS=0
  run "lanes3_conc2_tail4"          ZV_BENCH_LANES=3 ZV_TAIL_GROUPS=4
  run "lanes3_nolimit"              ZV_BENCH_LANES=3 ZV_MAX_CONCURRENT=0
  run "lanes4_conc2"                ZV_BENCH_LANES=4
  run "lanes4_conc3"                ZV_BENCH_LANES=4 ZV_MAX_CONCURRENT=3
  run "lanes2_nolimit (round 3)"    ZV_BENCH_LANES=2 ZV_MAX_CONCURRENT=0
done
