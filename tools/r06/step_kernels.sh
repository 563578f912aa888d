#!/bin/bash
# What does ONE replayed step launch?  Two rocprofv3 kernel traces of the default command, 20 and 220 timed steps (nothing else
# differs): the difference of the call counts / 200 is the per-step list, graph nodes included.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/stepk
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
F="--min-time 0 --warmup 5 --no-cpu-baseline --no-kernel-timers --no-other-precision"
for st in 20 220; do
  timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/s$st -- python3 $R/bench.py --steps $st $F > $O/s$st.log 2>&1
  echo "steps $st rc=$?"
done
python3 - <<PY
import csv, glob
def load(d, kind):
    f = glob.glob(f"$O/{d}/**/*{kind}_stats.csv", recursive=True)
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open(f[0]))} if f else {}
for kind in ("kernel", "memory_copy"):
    a, b = load("s20", kind), load("s220", kind)
    print(kind)
    for k in sorted(b, key=lambda k: -(b[k][0] - a.get(k, (0, 0))[0]) * b[k][1]):
        d = b[k][0] - a.get(k, (0, 0))[0]
        if d:
            print(f"  {d / 200:6.2f} per step x {b[k][1] / 1e3:8.1f} us  {k[:110]}")
PY
