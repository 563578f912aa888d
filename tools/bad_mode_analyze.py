"""Split a tools/bad_mode_trace.sh kernel trace into busy segments (between device syncs) and summarise each."""
import csv, glob, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = glob.glob(os.path.join(R, "gpurun_out", "badmode", "run", "*", "*kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
big = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], "enc" if "lstm_encode" in r["Kernel_Name"] else "dec")
       for r in rows if "lstm_encode" in r["Kernel_Name"] or "pointer_decode" in r["Kernel_Name"]]
segs, cur, last_end = [], [], None
for k in big:
    if last_end is not None and k[0] - last_end > 3_000_000:
        segs.append(cur); cur = []
    cur.append(k); last_end = max(last_end or 0, k[1])
segs.append(cur)
def ov(a, b): return max(0, min(a[1], b[1]) - max(a[0], b[0]))
for n, sg in enumerate(segs):
    if len(sg) < 80: continue
    sg = sg[len(sg) // 3:]
    enc = [k for k in sg if k[3] == "enc"]; dec = [k for k in sg if k[3] == "dec"]
    span = (max(k[1] for k in sg) - min(k[0] for k in sg)) / 1e6
    ee = sum(ov(a, b) for a in enc for b in enc if a is not b and a[2] != b[2]) / sum(a[1] - a[0] for a in enc)
    ed = sum(ov(a, b) for a in enc for b in dec if a[2] != b[2]) / sum(a[1] - a[0] for a in enc)
    dd = sum(ov(a, b) for a in dec for b in dec if a is not b and a[2] != b[2]) / sum(a[1] - a[0] for a in dec)
    print(f"segment {n}: {len(enc)} steps in {span:.1f} ms = {span / len(enc):.4f} ms/step; enc {sum(a[1]-a[0] for a in enc)/len(enc)/1e3:.0f} us "
          f"dec {sum(a[1]-a[0] for a in dec)/len(dec)/1e3:.0f} us; enc||enc {ee:.2f} enc||dec {ed:.2f} dec||dec {dd:.2f}")
