#!/bin/bash
# Round 6: first run of the two-buffer tiled aggregate: the aggregate tests, then both forms side by side.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06agg
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "tiled or tile_plan or aggregate" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python tools/bench_aggregate.py --configs ${CFG:-5000:128,2507:256,10000:32,5000:32} --forms tiled --tiled-forms 0,1 > $O/aggregate.jsonl 2> $O/aggregate.err; echo "bench rc=$?"
python3 - <<PY
import json
for l in open("$O/aggregate.jsonl"):
    r = json.loads(l)
    print(r["S"], r["copies"], {k: (v.get("ms"), v.get("ms_median"), v.get("frac_of_8TBps"), v["stream"]["efficiency"]) for k, v in r.items() if k.startswith("tiled")})
PY
tail -3 $O/aggregate.err
