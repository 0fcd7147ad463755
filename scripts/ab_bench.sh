# whole-step A/B of environment switches: bash scripts/ab_bench.sh tag "ENV=.." "ENV=.." ...   (ms_per_step of the pipelined bench)
O=gpurun_out/$1; shift
mkdir -p $O
for V in "$@"; do
  env $V python bench.py --no-cpu-baseline --no-extras > $O/b.json 2> $O/b.err || { echo "== $V FAILED"; tail -3 $O/b.err; continue; }
  python -c "
import json,sys; j=json.load(open('$O/b.json')); print('== $V', j['ms_per_step'], j['value'])" | tee -a $O/ab.txt
done
