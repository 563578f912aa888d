#!/bin/bash
# SQ counters of the split GIN layer kernels at the 1000-task shape (separate --pmc passes, kernel trace only).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04ginpmc; mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python3 $R/tools/bench_gin_layer.py --forms split --reps 2 > $O/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $O/$tag.log; }
done
python3 - <<'PY'
import csv, glob, os, collections, json
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r04ginpmc"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/*/*/*counter_collection.csv") + glob.glob(O + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gin_layer_split" not in k: continue
        key = "layer1" if "ILb1" in k or "<true" in k else "layer0"
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
