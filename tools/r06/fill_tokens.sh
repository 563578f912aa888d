#!/bin/bash
# Round 6: fill tokens of the tiled aggregate (at most `limit` of an XCD's 32 CUs in a fill at a time) — a sweep of the limit
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06tokens
mkdir -p $O
cd $R
for lim in ${LIMS:-0 8 12 16 20 24 0}; do
GNNPN_TILED_FILL_LIMIT=$lim timeout -k 10 300 python tools/bench_aggregate.py --configs ${CFG:-5000:128,2507:256,10000:32,20000:8} --forms tiled > $O/lim_$lim.jsonl 2> $O/lim_$lim.err; echo "limit $lim rc=$?"
python3 - <<PY
import json
for l in open("$O/lim_$lim.jsonl"):
    r = json.loads(l)
    t = r["tiled"]
    print("limit", $lim, r["S"], r["copies"], t.get("ms"), t.get("ms_best_round"), t.get("frac_of_8TBps"), t.get("bit_identical_to_gather"))
PY
done
