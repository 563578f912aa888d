#!/bin/bash
# What the RCCL path costs a rank at world size 1 (GNNPN_FORCE_DIST=1), by bucket size — one box, alternating fresh processes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06dist
mkdir -p $O
cd $R
for i in 1 2; do
for cfg in plain:0 dist:8 dist:32 dist:1; do
  mode=${cfg%%:*}; ge=${cfg#*:}
  extra=""; [ $mode = dist ] && extra="--gather-every $ge"
  fd=0; [ $mode = dist ] && fd=1
  GNNPN_FORCE_DIST=$fd timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision $extra > $O/${mode}_${ge}_$i.json 2> $O/${mode}_${ge}_$i.err
  python3 -c "
import json
d=json.loads([l for l in open('$O/${mode}_${ge}_$i.json') if l.startswith('{')][-1])
print('$mode', $ge, d['value'], d['ms_per_step'], d['per_rank'][0].get('collective_ms'), d['per_rank'][0].get('seats'))
"
done
done
