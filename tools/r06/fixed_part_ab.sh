#!/bin/bash
# tools/time_fixed_part.py (per-launch milliseconds of short encoder launches, back to back on one stream) in this tree and in another
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
for tree in ${ORDER:-. _b5 . _b5}; do
  cd $R/$tree
  echo "== $tree"
  timeout -k 10 300 python tools/time_fixed_part.py 2>/dev/null | cut -c1-400
done
