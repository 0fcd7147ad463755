# rocprofv3 kernel trace of the bench (eager launches) -> per-config summary on stdout
export TMPDIR=/tmp
rm -rf /tmp/zvprof && mkdir -p /tmp/zvprof
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zvprof -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-graph --no-extras --no-pipeline > /tmp/zvprof/bench.json 2>/tmp/zvprof/err.txt
python - <<'PY'
import csv, glob, collections, re
f = glob.glob('/tmp/zvprof/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'zv::' not in n: continue
    key = (re.sub(r'^(void )?zv::', '', n.split('(')[0]), r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r['LDS_Block_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'])
    a = agg.setdefault(key, [0,0]); a[0]+=1; a[1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
tot = sum(t for _,t in agg.values())
for k,(n,t) in agg.items():
    print(f"{k[0]:30s} grid=({k[1]},{k[2]},{k[3]}) lds={k[4]} vgpr={k[5]}+{k[6]} calls={n} avg_us={t/n/1000:.2f} share={t/tot:.3f}")
PY
