#!/bin/bash
# Round-6 measurement (after the epilogue rewrite) artefacts of the CSR aggregate: gpurun --timeout 900 -- 'bash tools/r06_aggregate_profiles.sh'
# Outputs under gpurun_out/r06aggprof/: the roofline table of all three forms, rocprofv3 kernel stats, and separate
# --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (with --kernel-trace only) of the tiled and the gather form.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06aggprof
mkdir -p $O
cd $R
CFG=${CFG:-2507:256,2507:64,5000:128,5000:32,10000:32,20000:8}
timeout -k 10 400 python tools/bench_aggregate.py --configs $CFG > $O/aggregate.jsonl 2> $O/aggregate.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
PCFG=${PCFG:-2507:256 5000:128 20000:8}
for cfg in $PCFG; do
  tag=$(echo $cfg | tr ':' 'x')
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -- python3 $R/tools/bench_aggregate.py --configs $cfg --forms gather,tiled --reps 20 > $O/stats_$tag.log 2>&1; echo "stats $tag rc=$?"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$tag -- python3 $R/tools/bench_aggregate.py --configs $cfg --forms gather,tiled --reps 2 > $O/fetch_$tag.log 2>&1; echo "fetch $tag rc=$?"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$tag -- python3 $R/tools/bench_aggregate.py --configs $cfg --forms gather,tiled --reps 2 > $O/write_$tag.log 2>&1; echo "write $tag rc=$?"
  for f in $(find $O/stats_$tag -name '*kernel_stats.csv'); do cp $f $O/kernel_stats_$tag.csv; done
done
find $O -name '*.db' -delete
python3 - <<PY
import csv, glob, json, collections, re
O = "$O"
out = {}
for cfg in "$PCFG".split():
    tag = cfg.replace(":", "x")
    def per_kernel(kind, counter):
        acc = collections.defaultdict(list)
        for f in glob.glob(f"{O}/{kind}_{tag}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                m = re.search(r"(csr_aggregate\w*kernel)", r["Kernel_Name"])
                if m and r["Counter_Name"] == counter:
                    acc[m.group(1)].append(float(r["Counter_Value"]))
        return acc
    fetch, write = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
    rec = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        fm, wm = (sum(f) / len(f) if f else None), (sum(w) / len(w) if w else None)
        rec[k] = {"launches": len(f), "FETCH_SIZE_KB": fm, "WRITE_SIZE_KB": wm,
                  "traffic": int((2 * fm + wm) * 1024) if fm is not None and wm is not None else None}
    out[tag] = rec
import sys
sys.path.insert(0, "$R")
from gnnpn_sc_amd._lib import source_hash
json.dump({"source_group": "aggregate", "source_hash": source_hash("aggregate"), "note": "separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (with --kernel-trace only) per shape S x copies; the counters are in KB; "
                   "traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch, FETCH doubled as the gfx950 guide prescribes; "
                   "Infinity-Cache hits are counted by FETCH_SIZE", "shapes": out}, open(O + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name '*kernel_trace.csv' -size +2M -delete
echo done
