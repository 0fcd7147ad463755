# serving-loop A/B on ONE box (interleaved rounds): batches in flight x utterance groups of the last vocoder stage
# (round 4 also tried a device-side limit of two concurrent batches with a third queued: no change, removed)
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
  run "lanes2 (default)"            ZV_BENCH_LANES=2
  run "lanes1"                      ZV_BENCH_LANES=1
  run "lanes3"                      ZV_BENCH_LANES=3
  run "lanes4"                      ZV_BENCH_LANES=4
  run "lanes2_tail1"                ZV_BENCH_LANES=2 ZV_TAIL_GROUPS=0
  run "lanes2_tail4"                ZV_BENCH_LANES=2 ZV_TAIL_GROUPS=4
  run "lanes2_tail16"               ZV_BENCH_LANES=2 ZV_TAIL_GROUPS=16
done
