#!/bin/bash
# the every-step interferer soak (tests/test_gpu_pipeline.py) on this tree and on the tree under ./_prev, alternating, one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do
for tree in new prev; do
  d=$R; [ $tree = prev ] && d=$R/_prev
  cd $d
  timeout -k 10 300 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "interferer and qws" > /tmp/t.log 2>&1; rc=$?
  python3 -c "
import json
d=json.load(open('$d/gpurun_out/parity/interferer_soak_qws_two_slots.json'))
print('$tree', 'rc=$rc', d['problems_per_s_without'], d['problems_per_s_with_interferer_every_step'], d['loss_pct_every_step'], d['loss_pct_every_8th_step'])
"
done
done
