#!/bin/bash
# RCCL path at world size 1: does the collective's footprint (channels = workgroups) explain its cost to the cooperative launches?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06dist
mkdir -p $O
cd $R
for nch in default 1 2 4; do
  if [ $nch = default ]; then unset NCCL_MAX_NCHANNELS; unset NCCL_MIN_NCHANNELS; else export NCCL_MAX_NCHANNELS=$nch; export NCCL_MIN_NCHANNELS=1; fi
  for ge in 8 1; do
  GNNPN_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision --gather-every $ge > $O/nch_${nch}_$ge.json 2> $O/nch_${nch}_$ge.err
  python3 -c "
import json
d=json.loads([l for l in open('$O/nch_${nch}_$ge.json') if l.startswith('{')][-1])
print('channels $nch gather-every $ge', d['value'], d['ms_per_step'], d['per_rank'][0].get('collective_ms'))
"
  done
done
