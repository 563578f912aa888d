#!/bin/bash
# The stagger probe (slot 1's first replay of every round held back by d microseconds) in this tree and in ./_prev (round 5's last commit)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/stagger
for tree in new prev new prev; do
  d=$R; [ $tree = prev ] && d=$R/_prev
  cd $d
  echo "== $tree"
  timeout -k 10 200 python tools/probes/stagger_probe.py --steps 100 --rounds 8 --delays ${DELAYS:-0,400} 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-700
done
