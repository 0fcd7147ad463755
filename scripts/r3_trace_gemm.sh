# durations of every conv_gemm_kernel / conv1d_mfma_kernel dispatch of ONE pass of the batch, in launch order
export TMPDIR=/tmp
rm -rf /tmp/zvkp
env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/zvkp -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-extras --no-pipeline > /tmp/zvkp_b.json 2>/tmp/zvkp_err.txt
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/zvkp/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'zv::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last pass: from the last embed kernel on
last = max(i for i, r in enumerate(rows) if 'embed_kernel' in r['Kernel_Name'])
for r in rows[last:]:
    n = r['Kernel_Name'].split('(')[0].replace('void ','').replace('zv::','')
    if 'conv' in n or 'norm_act' in n:
        print(f"{n[:44]:44s} grid=({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) us={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000:.1f}")
PY
