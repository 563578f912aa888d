#!/bin/bash
# All sixteen 4-byte positions of the encoder's step loop within a 64-byte line (trees _aN, N s_nop in front of the loop): solo launch
# time (tools/probes/dbg_solo_placement.py, settled launches) and the two-slot headline (the driver's command), twice each.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for n in 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15; do
  cd $R/_a$n
  solo=$(timeout -k 10 120 python $R/tools/probes/dbg_solo_placement.py 2>/dev/null | grep "^lds_kb 0" | sed 's/.*us per launch \[\([^]]*\)\].*/\1/' | awk -F', ' '{print $4, $5, $6}')
  head=$(timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision --no-kernel-timers 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'])")
  echo "nops $n: solo us $solo  headline $head"
done
done
