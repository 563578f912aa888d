#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06agg
mkdir -p $O
cd $R
for d in ${DBGS:-0 1 2}; do
GNNPN_TILED2_DBG=$d timeout -k 10 200 python tools/bench_aggregate.py --configs ${CFG:-5000:128,2507:256} --forms tiled --tiled-forms 1 --no-check > $O/dbg_$d.jsonl 2> $O/dbg_$d.err; echo "dbg $d rc=$?"
python3 - <<PY
import json
for l in open("$O/dbg_$d.jsonl"):
    r = json.loads(l)
    print("dbg", $d, r["S"], r["copies"], {k: (v.get("ms"), v.get("ms_median")) for k, v in r.items() if k.startswith("tiled")})
PY
done
