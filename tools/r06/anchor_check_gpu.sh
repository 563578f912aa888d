#!/bin/bash
# The anchored encoder loop (this tree) against un-anchored positions (_a0: 36 bytes into the line, _a2: 44, _a10: 12) and the pre-fix tree _b5
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for t in . _a0 _a2 _a10; do
  cd $R/$t
  solo=$(timeout -k 10 120 python $R/tools/probes/dbg_solo_placement.py 2>/dev/null | grep "^lds_kb 0" | sed 's/.*us per launch \[\([^]]*\)\].*/\1/' | awk -F', ' '{print $4, $5, $6}')
  head=$(timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], [(k['kernel'],k['avg_ms']) for k in d.get('kernels',[])][:2])")
  echo "$t: solo us $solo  headline $head"
done
done
