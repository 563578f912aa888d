#!/bin/bash
# The driver's command in several source trees, alternating fresh processes.  TREES="_a0 _a10 ." bash tools/r06/bench_trees.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do
for t in ${TREES:-. _b5}; do
  cd $R/$t
  echo -n "$t: "; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], [(k['kernel'],k['avg_ms']) for k in d.get('kernels',[])][:2])"
done
done
