"""Summarise the --pmc passes of tools/r05_recurrent_sq.sh: per (workload, arithmetic, kernel) the mean per dispatch of every
counter, and the derived figures DESIGN.md quotes.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* /
SQ_BUSY_CYCLES count quad-cycles (x 4 = cycles), SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs, GRBM_GUI_ACTIVE is
the sum over the 8 XCDs.
    python tools/r05_recurrent_sq_summary.py gpurun_out/r05sq   -> <dir>/summary.json (+ profiles/r05_recurrent_sq_summary.json when run in the repo)"""
import collections, csv, glob, json, os, sys
O = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05sq"
KERNELS = {"lstm_encode_coop_kernel": "lstm_encode_coop_kernel", "pointer_decode_lean_kernel": "pointer_decode_lean_kernel"}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(O, "*_*_[0-9]"))):
    wl, pr, _ = os.path.basename(d).rsplit("_", 2)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for k, short in KERNELS.items():
                if k in r["Kernel_Name"]:
                    agg[(wl, pr, short)][r["Counter_Name"]].append(float(r["Counter_Value"]))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd._lib import source_hash   # noqa: E402  (the tree the passes were measured on: summarise BEFORE editing csrc/)
out = {"source_group": "recurrent", "source_hash": source_hash("recurrent"), "method": "rocprofv3 --kernel-trace --pmc <3 counters per pass> -- python3 bench.py --workload W --precision P --graph 0 --inflight 1 "
                 "(eager, one stream, one launch at a time); mean per dispatch; tools/r05_recurrent_sq.sh",
       "units": "SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles as reported; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs; "
                "GRBM_GUI_ACTIVE summed over 8 XCDs", "kernels": {}}
for (wl, pr, k), c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    m["dispatches_seen"] = max(len(v) for v in c.values())
    g = m.get("GRBM_GUI_ACTIVE")
    d = {}
    if g:
        cyc = g / 8.0                                         # kernel duration in shader cycles
        d["kernel_cycles"] = round(cyc)
        simds = 1024.0                                        # 256 CUs x 4 SIMDs, one wavefront of the launch on each
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            d["mfma_busy_frac_of_all_simds"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (simds * cyc), 4)
        if "SQ_WAVE_CYCLES" in m:
            d["wave_resident_frac_of_simd_time"] = round(4 * m["SQ_WAVE_CYCLES"] / (simds * cyc), 4)
    w = m.get("SQ_WAVE_CYCLES")
    if w:
        for n, key in (("SQ_ACTIVE_INST_VALU", "valu_issue_frac_of_wave_life"), ("SQ_WAIT_INST_ANY", "waiting_on_instruction_frac_of_wave_life"),
                       ("SQ_WAIT_ANY", "waiting_any_frac_of_wave_life"), ("SQ_ACTIVE_INST_LDS", "lds_issue_frac_of_wave_life")):
            if n in m:
                d[key] = round(m[n] / w, 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            d["mfma_busy_frac_of_wave_life"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * w), 4)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac_of_lds_active"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
    out["kernels"][f"{wl}/{pr}/{k}"] = {"counters": {n: round(v, 1) for n, v in sorted(m.items())}, "derived": d}
json.dump(out, open(os.path.join(O, "summary.json"), "w"), indent=1)
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.isdir(os.path.join(R, "profiles")) and out["kernels"]:
    json.dump(out, open(os.path.join(R, "profiles", "r05_recurrent_sq_summary.json"), "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in out["kernels"].items()}, indent=1))
