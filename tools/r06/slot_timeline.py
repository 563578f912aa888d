"""Where are the two slots relative to each other?  Reads the rocprofv3 kernel trace of the default command (tools/r06/step_kernels.sh
leaves one under gpurun_out/stepk/s220/) and reports, over a window of settled steps: the share of time with 0 / 1 / 2 recurrent
kernels running, the step period, and the first steps' timeline (start, end, duration in us, kernel, hardware queue).
    python tools/r06/slot_timeline.py [gpurun_out/stepk/s220]"""
import collections, csv, glob, json, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stepk/s220"
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
NAMES = (("lstm_encode_coop_kernel<2", "ENC"), ("pointer_decode_lean_kernel<true, 2", "DEC"), ("gin_request", "gin"), ("linear_f32", "lin"),
         ("coop_zero", "zero"), ("select_cand", "sel"), ("pick_prob", "pick"), ("qos_reward", "qos"), ("copyBuffer", "copy"))
def short(n):
    return next((s for k, s in NAMES if k in n), None)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"])
            for r in csv.DictReader(open(f)) if short(r["Kernel_Name"]))
queues = [q for q, _ in collections.Counter(e[3] for e in ev if e[2] == "ENC").most_common(2)]      # the two slots' hardware queues
enc = [e for e in ev if e[2] == "ENC" and e[3] in queues]
lo, hi = len(enc) // 4, 3 * len(enc) // 4
t0, t1 = enc[lo][0], enc[hi][0]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
pts = sorted(p for s, e, k, q in win if k in ("ENC", "DEC") for p in ((s, 1), (e, -1)))
cur, last, acc = 0, t0, collections.Counter()
for t, dlt in pts:
    acc[cur] += t - last
    last, cur = t, cur + dlt
acc[cur] += t1 - last
tot = sum(acc.values())
dur = collections.defaultdict(list)
for s, e, k, q in win:
    dur[k].append((e - s) / 1e3)
gap = []                                                     # a slot's last kernel of a step (qos) -> its next step's first (copy)
for q in queues:
    mine = [e for e in win if e[3] == q]
    gap += [(b[0] - a[1]) / 1e3 for a, b in zip(mine, mine[1:]) if a[2] == "qos" and b[2] == "copy"]
period = 2 * (t1 - t0) / 1e3 / (hi - lo)                     # a slot's step period: two slots alternate
offs = [min(o, period - o) for o in (abs(a[0] - b[0]) / 1e3 for a, b in zip(enc[lo:hi], enc[lo + 1:hi + 1]) if a[3] != b[3])]
print(json.dumps({"trace": f, "steps_in_window": hi - lo, "ms_per_step": round((t1 - t0) / 1e6 / (hi - lo), 4),
                  "share_of_time_with_n_recurrent_kernels": {str(k): round(v / tot, 4) for k, v in sorted(acc.items())},
                  "mean_us": {k: round(sum(v) / len(v), 1) for k, v in dur.items()},
                  "encoder_start_offset_between_slots_us": {"median": round(sorted(offs)[len(offs) // 2], 1), "min": round(min(offs), 1), "max": round(max(offs), 1)},
                  "gap_between_a_slots_replays_us": round(sum(gap) / max(1, len(gap)), 1)}))
base = win[0][0]
for s, e, k, q in win[:24]:
    print(f"{(s - base) / 1e3:9.1f} {(e - base) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {k:5s} q{q}")
