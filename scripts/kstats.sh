#!/bin/bash
# Resource usage of the device kernels (registers, spills, scratch, LDS, occupancy) as the compiler reports it for the flags of
# nightmare_rl_amd/csrc/Makefile.   usage: scripts/kstats.sh [name filter]   (EXTRA="-D..." scripts/kstats.sh for variants)
make -s -C "$(dirname "$0")/../nightmare_rl_amd/csrc" kstats EXTRA="$EXTRA" | python3 -c "
import sys
pat = sys.argv[1] if len(sys.argv) > 1 else ''
cur, out = None, []
for line in sys.stdin:
    t = line.strip()
    if t.startswith('Function Name:'):
        cur, out = t.split(':', 1)[1].strip(), []
    elif cur is not None:
        out.append(t)
        if t.startswith('LDS Size'):
            if pat in cur: print(cur[:64], '|', '; '.join(x for x in out if not x.startswith('Dynamic')))
            cur = None
" "$1"
