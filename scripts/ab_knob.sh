# one-box A/B of library switches: interleaved rounds of bench.py per arm
# usage (GPU box): bash scripts/ab_knob.sh <rounds> "NAME=VAL ..." "NAME=VAL ..." ...   ("-" = defaults)
R=$1; shift
for r in $(seq $R); do
  for arm in "$@"; do
    envs=""; [ "$arm" != "-" ] && envs="$arm"
    env $envs python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st={s['stage']:s['ms'] for s in d['roofline']['stages']}
k={x['name']:x['ms_per_step'] for x in d['extra']['kernels']}
print('%-40s ms_per_step %.3f  resblock %s = %.3f  dec_conv %.3f  upsample %.3f' % ('$arm', d['ms_per_step'], ' '.join('%s %.3f'%(a,b) for a,b in sorted(st.items())), d['roofline']['ms_per_pass'], k.get('dec_conv',0), k.get('voc_upsample',0)))"
  done
done
