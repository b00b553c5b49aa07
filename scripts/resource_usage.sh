#!/bin/bash
# usage: scripts/resource_usage.sh [extra hipcc flags]  -> VGPR / spill / LDS / occupancy of every kernel in lfr_solve.hip (runs here, no GPU)
R=$(cd $(dirname $0)/.. && pwd); C=$R/local-feature-refinement_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I $R/include -I $C "$@" -c $C/lfr_solve.hip -o /tmp/lfr_solve_ru.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r'remark:\s+(.*?)\s+\[-Rpass', line)
    if not m: continue
    t = m.group(1)
    if t.startswith('Function Name:'):
        cur = t.split(':', 1)[1].strip(); rows[cur] = {}
    elif cur and ':' in t:
        k, v = t.split(':', 1); rows[cur][k.strip()] = v.strip()
import subprocess
for f, d in rows.items():
    if 'rocprim' in f or 'hipcub' in f: continue
    name = subprocess.run(['c++filt', f], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(anonymous namespace\)::', '', name)[:70]
    print('%-72s VGPR %4s AGPR %4s spill %4s scratch %5s LDS %6s occ %s' % (name, d.get('VGPRs'), d.get('AGPRs'), d.get('VGPRs Spill'), d.get('ScratchSize [bytes/lane]'), d.get('LDS Size [bytes/block]'), d.get('Occupancy [waves/SIMD]')))
"
