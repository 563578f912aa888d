#!/bin/bash
# Round 5: SQ counters of csr_aggregate_tiled_kernel (separate --pmc passes with the kernel trace only) over tools/bench_aggregate.py:
#     gpurun --timeout 900 -- 'bash tools/r05_aggregate_sq.sh'   -> gpurun_out/r05aggsq/summary.json (copy: profiles/r05_aggregate_sq_summary.json)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05aggsq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY"
      "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM")
for cfg in ${CFGS:-5000:128 2507:256 20000:8}; do
  tag0=$(echo $cfg | tr ':' 'x'); i=0
  for set in "${SETS[@]}"; do
    tag=${tag0}_$i; i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python3 $R/tools/bench_aggregate.py --configs $cfg --forms tiled --reps 2 > $O/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $O/$tag.log; }
    echo "pass $tag done"
  done
done
find $O -name '*kernel_trace.csv' -size +4M -delete
find $O -name '*.db' -delete
python3 - <<PY
import collections, csv, glob, json, os
O = "$O"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(O, "*x*_[0-9]"))):
    shape = os.path.basename(d).rsplit("_", 1)[0]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "csr_aggregate_tiled_kernel" in r["Kernel_Name"]:
                agg[shape][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"method": "rocprofv3 --kernel-trace --pmc <3 counters per pass> -- python3 tools/bench_aggregate.py --configs S:copies --forms tiled; mean per dispatch of csr_aggregate_tiled_kernel",
       "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES in quad-cycles as reported; GRBM_GUI_ACTIVE summed over 8 XCDs", "shapes": {}}
for shape, c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    d = {}
    g = m.get("GRBM_GUI_ACTIVE")
    w = m.get("SQ_WAVE_CYCLES")
    if g:
        cyc = g / 8.0
        d["kernel_cycles"] = round(cyc)
        if w: d["wave_resident_frac_of_simd_time (16 waves per CU = 4 per SIMD -> 4.0 when always resident)"] = round(4 * w / (1024.0 * cyc), 3)
        if "SQ_ACTIVE_INST_VALU" in m: d["valu_issue_busy_frac_of_simd_time"] = round(4 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc), 4)
        if "SQ_ACTIVE_INST_LDS" in m: d["lds_issue_busy_frac_of_simd_time"] = round(4 * m["SQ_ACTIVE_INST_LDS"] / (1024.0 * cyc), 4)
        if "SQ_LDS_IDX_ACTIVE" in m: d["lds_array_active_frac_of_cu_time"] = round(m["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc), 4)
    if w:
        for n, key in (("SQ_WAIT_INST_ANY", "waiting_on_instruction_frac_of_wave_life"), ("SQ_WAIT_ANY", "waiting_any_frac_of_wave_life"), ("SQ_WAIT_INST_LDS", "waiting_on_lds_frac_of_wave_life")):
            if n in m: d[key] = round(m[n] / w, 4)
    if m.get("SQ_LDS_IDX_ACTIVE"): d["lds_conflict_cycles_over_active"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"], 4)
    waves = m.get("SQ_WAVES")
    if waves:
        for n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if n in m: d[n.lower() + "_per_wave"] = round(m[n] / waves, 1)
    out["shapes"][shape] = {"counters": {k: round(v, 1) for k, v in m.items()}, "derived": d}
json.dump(out, open(os.path.join(O, "summary.json"), "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in out["shapes"].items()}, indent=1))
PY
echo done
