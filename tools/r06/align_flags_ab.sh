#!/bin/bash
# Compiler-wide alignment flags against the hand anchor: -falign-loops=64 / 32, -mllvm -align-all-nofallthru-blocks=5 (trees _vL64, _vL32, _vNF5)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for t in ${TREES:-. _vL64 _vL32 _vNF5}; do
  cd $R/$t
  solo=$(timeout -k 10 120 python $R/tools/probes/dbg_solo_placement.py 2>/dev/null | grep "^lds_kb 0" | sed 's/.*us per launch \[\([^]]*\)\].*/\1/' | awk -F', ' '{print $4, $5, $6}')
  head=$(timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], [(k['kernel'],k['avg_ms']) for k in d.get('kernels',[])][:2])")
  echo "$t: solo us $solo  headline $head"
done
done
