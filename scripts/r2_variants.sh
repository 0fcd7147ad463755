# per-kernel times of the batch workload under tile-shape overrides: bash scripts/r2_variants.sh tag "ENV=.. ENV=.." ...
export TMPDIR=/tmp
O=gpurun_out/$1; shift
mkdir -p $O
for V in "$@"; do
  echo "== $V" >> $O/variants.txt
  rm -rf /tmp/zvkp
  env $V rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zvkp -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-extras --no-pipeline > $O/vb.json 2>$O/vb.err || { tail -5 $O/vb.err >> $O/variants.txt; continue; }
  python - >> $O/variants.txt <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/zvkp/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'zv::' not in n: continue
    key = (n.split('(')[0].replace('void ','').replace('zv::',''), r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    a = agg.setdefault(key, [0,0]); a[0]+=1; a[1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
tot = sum(t for _,t in agg.values())
for k,(n,t) in agg.items():
    if t/tot > 0.008: print(f"{k[0]:34s} grid=({k[1]},{k[2]},{k[3]}) calls={n} avg_us={t/n/1000:.1f} share={t/tot:.3f}")
print("total ms per step", tot/1e6/8)
PY
done
cat $O/variants.txt
