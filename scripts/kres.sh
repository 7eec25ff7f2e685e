#!/bin/bash
# per-kernel register / spill / occupancy table of one .hip file (compiler view):  scripts/kres.sh audiolm-pytorch_amd/csrc/attention.hip [filter]
f=$1; filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I"$(dirname $f)" -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re, subprocess
cur = None; rows = {}
for ln in sys.stdin:
    m = re.search(r'remark:\s+(Function Name|VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\S+)', ln)
    if not m: continue
    k, v = m.groups()
    if k == 'Function Name':
        cur = subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '').split('(')[0]
        rows[cur] = {}
    elif cur: rows[cur][k] = v
for n, r in rows.items():
    print(f'{n[:70]:70s} vgpr {r.get(\"VGPRs\"):>4s} agpr {r.get(\"AGPRs\"):>4s} spill {r.get(\"VGPRs Spill\"):>4s} scratch {r.get(\"ScratchSize [bytes/lane]\"):>5s} occ {r.get(\"Occupancy [waves/SIMD]\")}')
" | grep -E "$filt"
