# rocprofv3 PMC passes (counters only, no tracing flags besides kernel-trace) -> per-kernel averages
# usage: bash scripts/pmc.sh "<counters...>" [env...]
export TMPDIR=/tmp
CNT="$1"; shift
rm -rf /tmp/zvpmc && mkdir -p /tmp/zvpmc
env "$@" rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d /tmp/zvpmc -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-graph --no-extras --no-pipeline > /tmp/zvpmc/bench.json 2>/tmp/zvpmc/err.txt
python - <<'PY'
import csv, glob, collections, re
fs = glob.glob('/tmp/zvpmc/**/*counter_collection.csv', recursive=True)
if not fs:
    print(open('/tmp/zvpmc/err.txt').read()[-2000:]); raise SystemExit
agg = collections.OrderedDict()
for r in csv.DictReader(open(fs[0])):
    n = r['Kernel_Name']
    if 'zv::' not in n: continue
    key = (re.sub(r'^(void )?zv::', '', n.split('(')[0]), r['Grid_Size'], r.get('LDS_Block_Size',''))
    d = agg.setdefault(key, collections.OrderedDict())
    c = d.setdefault(r['Counter_Name'], [0, 0.0]); c[0]+=1; c[1]+=float(r['Counter_Value'])
for k, d in agg.items():
    print(k[0], 'grid', k[1], ' '.join(f"{c}={v/n:.4g}" for c,(n,v) in d.items()), 'n=%d' % list(d.values())[0][0])
PY
