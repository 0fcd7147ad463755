# rocprofv3 kernel trace of any python script of this repo -> per-kernel-configuration summary on stdout
# usage: [ENV=..] bash scripts/kprof.sh scripts/vocT.py 8192
export TMPDIR=/tmp
D=/tmp/zvkprof
rm -rf $D && mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python "$@" > $D/out.txt 2>$D/err.txt || { tail -20 $D/err.txt; exit 1; }
cat $D/out.txt
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/zvkprof/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'zv::' not in n: continue
    key = (n.split('(')[0].replace('void ','').replace('zv::',''), r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r['LDS_Block_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'])
    a = agg.setdefault(key, [0,0]); a[0]+=1; a[1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
tot = sum(t for _,t in agg.values())
for k,(n,t) in agg.items():
    print(f"{k[0]:34s} grid=({k[1]},{k[2]},{k[3]}) lds={k[4]} vgpr={k[5]}+{k[6]} calls={n} avg_us={t/n/1000:.2f} share={t/tot:.3f}")
PY
