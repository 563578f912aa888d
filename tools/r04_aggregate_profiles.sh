#!/bin/bash
# Round-4 measurement artefacts of the CSR aggregate: gpurun --timeout 900 -- 'bash tools/r04_aggregate_profiles.sh'
# Outputs under gpurun_out/r04aggprof/: the roofline table of all three forms, rocprofv3 kernel stats, and separate
# --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (with --kernel-trace only) of the tiled and the gather form.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04aggprof
mkdir -p $O
cd $R
CFG=${CFG:-2507:256,2507:64,5000:128,5000:32,10000:32,20000:8}
timeout -k 10 400 python tools/bench_aggregate.py --configs $CFG > $O/aggregate.jsonl 2> $O/aggregate.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
PCFG=${PCFG:-2507:256,5000:128,20000:8}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/bench_aggregate.py --configs $PCFG --forms gather,tiled --reps 5 > $O/stats.log 2>&1; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/bench_aggregate.py --configs $PCFG --forms gather,tiled --reps 2 > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/bench_aggregate.py --configs $PCFG --forms gather,tiled --reps 2 > $O/write.log 2>&1; echo "write rc=$?"
find $O -name '*.db' -delete
python3 - <<PY
import csv, glob, json, collections
O = "$O"
cfgs = [tuple(int(v) for v in c.split(":")) for c in "$PCFG".split(",")]
def per_kernel(pattern, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(O + "/" + pattern + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("aggregate" in r["Kernel_Name"]):
                out[(r["Kernel_Name"].split("(")[0].split("<")[0].split(" ")[-1], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return out
fetch, write = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
rows = {}
for key in sorted(set(fetch) | set(write)):
    f = fetch.get(key, []); w = write.get(key, [])
    fm = sum(f) / len(f) if f else None; wm = sum(w) / len(w) if w else None
    rows[f"{key[0]} grid {key[1]}"] = {"launches": len(f), "FETCH_SIZE_KB": fm, "WRITE_SIZE_KB": wm,
        "traffic_bytes_fetch_doubled": (2 * fm + wm) * 1024 if fm is not None and wm is not None else None}
json.dump({"configs": cfgs, "note": "separate --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH doubled as the gfx950 guide prescribes (counter in KB); keyed by kernel and grid size (threads)", "kernels": rows}, open(O + "/pmc_by_kernel.json", "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
for f in $(find $O/stats -name '*kernel_stats.csv'); do cp $f $O/kernel_stats.csv; done
find $O -name '*kernel_trace.csv' -size +2M -delete
echo done
