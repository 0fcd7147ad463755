# how much do the kernels of the two lanes overlap in time?  kernel trace of the pipelined, graph-replayed bench:
# sum of kernel durations vs the union of their intervals, and the busiest overlapping pairs.  usage: bash scripts/overlap.sh [ENV=..]
export TMPDIR=/tmp
D=/tmp/zvovl
rm -rf $D && mkdir -p $D
env "$@" rocprofv3 --kernel-trace --output-format csv -d $D -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $D/out.json 2>$D/err.txt || { tail -5 $D/err.txt; exit 1; }
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/zvovl/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'zv::' not in n: continue
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n.split('(')[0].replace('void ','').replace('zv::','')[:40], r.get('Queue_Id','?')))
rows.sort()
# keep the last 60 % (steady state)
t0 = rows[int(len(rows)*0.4)][0]
rows = [r for r in rows if r[0] >= t0]
tot = sum(e - s for s, e, _, _ in rows)
# union
cur_s, cur_e, uni = rows[0][0], rows[0][1], 0
for s, e, _, _ in rows[1:]:
    if s > cur_e: uni += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
uni += cur_e - cur_s
span = max(e for _, e, _, _ in rows) - rows[0][0]
print(f"kernels {len(rows)}  sum of durations {tot/1e6:.2f} ms  union {uni/1e6:.2f} ms  span {span/1e6:.2f} ms  queues {sorted(set(q for *_, q in rows))}")
# overlapped time per kernel name
ov = collections.Counter(); dur = collections.Counter(); cnt = collections.Counter()
active = []
for i, (s, e, n, q) in enumerate(rows):
    dur[n] += e - s; cnt[n] += 1
    for (s2, e2, n2, q2) in rows[max(0, i - 6):i]:
        o = min(e, e2) - max(s, s2)
        if o > 0 and q2 != q: ov[n] += o; ov[n2] += o
for n, d in dur.most_common(14):
    print(f"  {n:42s} calls {cnt[n]:4d} avg {d/cnt[n]/1e3:8.1f} us  overlapped with the other queue {ov[n]/max(d,1):.2f}")
PY
python -c "import json; j=json.load(open('/tmp/zvovl/out.json')); print('ms_per_step', j['ms_per_step'])"
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/zvovl/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n.split('(')[0].replace('void ','').replace('zv::','')[:34], r.get('Queue_Id','?')))
rows.sort()
t0 = rows[int(len(rows)*0.4)][0]
rows = [r for r in rows if r[0] >= t0]
gaps = []
cur_e, last = rows[0][1], rows[0]
for r in rows[1:]:
    if r[0] > cur_e:
        gaps.append((r[0] - cur_e, last[2], last[3], r[2], r[3], (cur_e - t0) / 1e6))
    if r[1] > cur_e: cur_e, last = r[1], r
gaps.sort(reverse=True)
print("idle gaps: total %.2f ms in %d gaps; > 20 us: %.2f ms in %d" % (sum(g[0] for g in gaps) / 1e6, len(gaps), sum(g[0] for g in gaps if g[0] > 20000) / 1e6, sum(1 for g in gaps if g[0] > 20000)))
for g in gaps[:25]:
    print(f"  {g[0]/1e3:8.1f} us at {g[5]:7.2f} ms  after {g[1]:34s} q{g[2]}  before {g[3]:34s} q{g[4]}")
PY
