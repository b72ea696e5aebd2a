#!/bin/bash
# VGPR / SGPR / LDS / scratch / occupancy of every kernel in a HIP source: tools/kernel_resources.sh <file.hip> [filter]
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c -o /dev/null "$1" -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r'remark: [^ ]+ (Function Name|    [A-Za-z ]+\[?[a-z/ ]*\]?): (.*) \[-Rpass', line) or re.search(r'remark: (?:[^ ]+ )?\s*([A-Za-z][A-Za-z \[\]/]*): (\S+)', line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k in ('Function Name', 'Name'):
        if cur: rows.append(cur)
        cur = {'name': v}
    else: cur[k] = v
if cur: rows.append(cur)
flt = sys.argv[1] if len(sys.argv) > 1 else ''
names = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    if 'rocprim' in n or flt not in n: continue
    print(f\"{n[:60]:60s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} SGPR {r.get('TotalSGPRs','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>3s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} LDS {r.get('LDS Size [bytes/block]','?')}\")
" "$2"
