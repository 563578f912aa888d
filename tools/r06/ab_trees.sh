#!/bin/bash
# Same-box A/B of two source trees: this one against a worktree of an earlier commit at ./_prev (built in-tree, travels with the snapshot).
# Alternating fresh processes of the driver's command.   gpurun -- 'bash tools/r06/ab_trees.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06ab
mkdir -p $O
for i in 1 2 3 4; do
for tree in new prev; do
  d=$R; [ $tree = prev ] && d=$R/${PREV:-_prev}
  cd $d
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision > $O/${tree}_$i.json 2> $O/${tree}_$i.err
  python3 -c "
import json
d=json.loads([l for l in open('$O/${tree}_$i.json') if l.startswith('{')][-1])
print('$tree', d['value'], d['ms_per_step'], [ (k['kernel'],k['avg_ms']) for k in d.get('kernels',[])][:2])
"
done
done
