#!/bin/bash
# The stagger probe over several source trees (./_prev, ./_b1 ...: git archive of a commit, built in-tree): which commit made a
# staggered start of the two slots end in hand-off time-outs?     TREES="_prev _b1 _b2 . " bash tools/r06/stagger_bisect.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
for tree in ${TREES:-_prev _b1 _b2 _b3 _b4 .}; do
  cd $R/$tree
  echo "== $tree"
  timeout -k 10 240 python tools/probes/stagger_probe.py --steps 100 --rounds ${ROUNDS:-12} --delays ${DELAYS:-400} 2>&1 | grep "^delay\|FAILED\|GnnpnError" | cut -c1-400
done
