#!/bin/bash
# Round 6, first call: (1) LDS-DMA above 64 KB probe, (2) the three HIP-event instruments of tools/time_instruments.py plain and
# under rocprofv3 --kernel-trace --stats.  gpurun --timeout 600 -- 'bash tools/r06/instruments.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06instr
mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_dma_high_probe tools/probes/lds_dma_high_probe.hip && timeout -k 10 60 /tmp/lds_dma_high_probe > $O/lds_dma_high_probe.json 2>&1; echo "probe rc=$?"
cat $O/lds_dma_high_probe.json
FORM=${FORM:-default}
for cfg in 5000:128 2507:256 20000:8; do
  tag=$(echo $cfg | tr ':' 'x')
  timeout -k 10 200 python tools/time_instruments.py --config $cfg --form $FORM > $O/plain_$tag.json 2> $O/plain_$tag.err; echo "plain $tag rc=$?"
done
cd /tmp && export TMPDIR=/tmp
for cfg in 5000:128 2507:256 20000:8; do
  tag=$(echo $cfg | tr ':' 'x')
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 $R/tools/time_instruments.py --config $cfg --form $FORM > $O/prof_$tag.json 2> $O/prof_$tag.err; echo "prof $tag rc=$?"
  for f in $(find $O/prof_$tag -name '*kernel_stats.csv'); do cp $f $O/kernel_stats_$tag.csv; done
  # per-dispatch durations of the aggregate in launch order (the trace), reduced to a short summary
  python3 - <<PY
import csv, glob, json, statistics
rows = []
for f in glob.glob("$O/prof_$tag/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csr_aggregate_tiled" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
d = [(e - s) / 1e6 for s, e in rows]
gaps = [(rows[i + 1][0] - rows[i][1]) / 1e6 for i in range(len(rows) - 1)]
starts = [(rows[i + 1][0] - rows[i][0]) / 1e6 for i in range(len(rows) - 1)]
b2b = [s for s, g in zip(starts, gaps) if g < 0.05]
json.dump({"dispatches": len(d), "duration_ms": {"mean": statistics.mean(d), "median": statistics.median(d), "min": min(d), "max": max(d)},
           "start_to_start_ms_in_trains": {"n": len(b2b), "median": statistics.median(b2b) if b2b else None, "mean": statistics.mean(b2b) if b2b else None},
           "gap_ms_in_trains_median": statistics.median([g for g in gaps if g < 0.05]) if b2b else None,
           "negative_gaps": sum(1 for g in gaps if g < 0)}, open("$O/trace_summary_$tag.json", "w"), indent=1)
PY
  cat $O/trace_summary_$tag.json
done
find $O -name '*.db' -delete
find $O -name '*kernel_trace.csv' -size +1M -delete
cat $O/plain_*.json $O/prof_*.json
echo done
