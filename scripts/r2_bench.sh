# bench + per-kernel trace of the batch workload
export TMPDIR=/tmp
O=gpurun_out/${1:-r2b}
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - <<PY
import json
j=json.load(open("$O/bench.json"))
print("VALUE", j["value"], "ms/step", j["ms_per_step"], "roofline frac", j["roofline"]["frac"], "TF", j["roofline"]["mfma_TFLOPs"])
print("cpu", j["cpu_baseline"])
for k in j["extra"]["kernels"]: print("  %-22s n=%3d avg %9.1f us  ms/step %8.3f share %.3f  %7.1f GB/s %7.1f TF" % (k["name"],k["launches_per_step"],k["avg_us"],k["ms_per_step"],k["share"],k["algo_GBps"],k["algo_TFLOPs"]))
print({k:v for k,v in j["extra"].items() if k!="kernels"})
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zvkp -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-extras > $O/prof_bench.json 2>$O/prof.err
python - <<'PY' > $O/kernel_trace_summary.txt
import csv, glob, collections
f = glob.glob('/tmp/zvkp/**/*kernel_trace.csv', recursive=True)[0]
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
cat $O/kernel_trace_summary.txt
