#!/bin/bash
# Round-4 diagnosis of the tiled aggregate: gpurun --timeout 900 -- 'bash tools/r04_aggregate_probe.sh'.  Outputs: gpurun_out/r04agg/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04agg
mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -k "tiled" > $O/pytest_tiled.log 2>&1; echo "tiled tests rc=$?"; tail -3 $O/pytest_tiled.log
timeout -k 10 300 python tools/bench_aggregate.py --configs ${CFG:-2507:256,5000:128,10000:32,20000:8} > $O/aggregate.jsonl 2> $O/aggregate.err; echo "bench rc=$?"
python3 - <<PY
import json
for ln in open("$O/aggregate.jsonl"):
    d = json.loads(ln)
    f = lambda k: (d.get(k, {}).get("ms"), d.get(k, {}).get("frac_of_8TBps"))
    print(d["S"], d["copies"], "gather", f("gather"), "lds", f("lds"), "tiled", f("tiled"), d.get("tiled", {}).get("stream", {}).get("efficiency"))
PY
timeout -k 10 500 python tools/ablate_aggregate.py run ${CFGS:-2507:256 5000:128 20000:8} > $O/ablate.jsonl 2> $O/ablate.err; echo "ablate rc=$?"
cat $O/ablate.jsonl
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -- python3 $R/tools/bench_aggregate.py --configs 5000:128 --forms tiled --reps 3 > $O/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "tiled_kernel" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        print(f.split("/")[-3] if "/" in f else f, k, "launches", n, "mean", v / max(n, 1))
PY
find $O -name '*.db' -delete; find $O -name '*kernel_trace.csv' -size +2M -delete
echo done
