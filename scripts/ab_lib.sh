# same-box A/B of two builds of the library (interleaved rounds): the in-tree build against another .so (ZEROVOX_AMD_LIB)
# usage (GPU box): bash scripts/ab_lib.sh zerovox.cpp_amd/_ab/libzv_X.so [rounds]  -> per arm and round: ms_per_step and the stage times
O=$1; R=${2:-3}
run() {   # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s ms_per_step %.3f  ' % ('$name', d['ms_per_step']) + '  '.join('%s %.3f' % (s['stage'], s['ms']) for s in d['roofline']['stages']))"
}
for r in $(seq $R); do
  run other ZEROVOX_AMD_LIB=$O
  run tree  ZV_NOP=0
done
