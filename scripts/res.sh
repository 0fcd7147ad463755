# per-kernel register / scratch / occupancy summary of a HIP source (hipcc -Rpass-analysis=kernel-resource-usage)
# usage: bash scripts/res.sh zerovox.cpp_amd/csrc/conv1d_mfma.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-value $EXTRA -c $1 -o /tmp/res_tmp.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur={}
for line in sys.stdin:
    if 'error' in line or 'warning' in line: print(line.rstrip())
    m=re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m: continue
    t=m.group(1)
    if t.startswith('Function Name:'):
        cur={'name':t.split(':',1)[1].strip()}
    else:
        k,v=t.split(':',1); cur[k.strip()]=v.strip()
        if k.strip().startswith('LDS Size'):
            n=subprocess.run(['c++filt',cur['name']],capture_output=True,text=True).stdout.strip()
            n=n.replace('zv::','').split('(')[0].replace('void ','')
            print(f\"{n:44s} vgpr {cur.get('VGPRs','?'):>4} agpr {cur.get('AGPRs','?'):>3} sgpr {cur.get('SGPRs','?'):>3} scratch {cur.get('ScratchSize [bytes/lane]','?'):>3} occ {cur.get('Occupancy [waves/SIMD]','?')}\")
"
