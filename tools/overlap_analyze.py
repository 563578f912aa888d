"""Per-run summary of tools/overlap_modes.sh traces: durations and how the two slots' kernels overlap."""
import csv, glob, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in sorted(glob.glob(os.path.join(R, "gpurun_out", "modes", "run*"))):
    if not os.path.isdir(d): continue
    f = glob.glob(d + "/*/*kernel_trace.csv")
    if not f: continue
    rows = list(csv.DictReader(open(f[0])))
    val = open(d + ".value.txt").read()
    try: v = json.loads(open(d + ".log").read().strip().splitlines()[-1])["value"]
    except Exception: v = val[:80]
    enc = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows if "lstm_encode" in r["Kernel_Name"])
    dec = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows if "pointer_decode" in r["Kernel_Name"])
    enc, dec = enc[len(enc) // 2:], dec[len(dec) // 2:]            # steady state: second half
    def ov(a, b): return max(0, min(a[1], b[1]) - max(a[0], b[0]))
    ee = sum(ov(e, o) for e in enc for o in enc if o is not e and o[2] != e[2]) / max(1, sum(e[1] - e[0] for e in enc))
    ed = sum(ov(e, o) for e in enc for o in dec if o[2] != e[2]) / max(1, sum(e[1] - e[0] for e in enc))
    dd = sum(ov(e, o) for e in dec for o in dec if o is not e and o[2] != e[2]) / max(1, sum(e[1] - e[0] for e in dec))
    print(os.path.basename(d), "value", v, "enc avg us %.0f" % (sum(e[1] - e[0] for e in enc) / len(enc) / 1e3),
          "dec avg us %.0f" % (sum(e[1] - e[0] for e in dec) / len(dec) / 1e3),
          "enc||enc %.2f enc||dec %.2f dec||dec %.2f" % (ee, ed, dd))
