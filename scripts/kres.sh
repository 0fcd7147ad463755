# per-kernel register / spill / occupancy report of a .hip file: bash scripts/kres.sh conv1d_mfma.hip [substring of the kernel name]
cd "$(dirname "$0")/../zerovox.cpp_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $EXTRA -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 |
python3 -c "
import re, sys, subprocess
pat = sys.argv[1] if len(sys.argv) > 1 else ''
cur = None; rows = {}
for ln in sys.stdin:
    m = re.search(r'Function Name: (\S+)', ln)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass', ln)
    if m and cur: rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip().replace('void ','').replace('zv::','')
    if pat in name: print(f\"{name.split('(')[0]:58s} vgpr {v.get('VGPRs','?'):>4s} agpr {v.get('AGPRs','?'):>4s} sgpr {v.get('TotalSGPRs','?'):>4s} scratch {v.get('ScratchSize','?'):>4s} spill {v.get('VGPRs Spill','?'):>3s} occ {v.get('Occupancy','?')}\")
" "$2"
