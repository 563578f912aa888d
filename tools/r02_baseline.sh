#!/bin/bash
# Round-2 baseline measurements of the round-1 kernels at the shapes the verdict asked for (working set > 256 MB
# Infinity Cache): gpurun --timeout 1500 -- 'bash tools/r02_baseline.sh'.  Outputs: gpurun_out/r02base/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02base
mkdir -p $O
cd $R
timeout 400 python bench.py --workload synth5 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_synth5.json 2> $O/bench_synth5.err
timeout 400 python bench.py --workload synth5 --precision f16 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_synth5_f16.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
for wl in normal synth4; do
  st=8; [ $wl = synth4 ] && st=3
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python3 $R/bench.py --workload $wl --steps $st --warmup 1 --no-cpu-baseline --no-kernel-timers --no-split-line --graph 0 --inflight 1 > $O/stats_$wl.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$wl -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-split-line --graph 0 --inflight 1 > $O/fetch_$wl.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$wl -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-split-line --graph 0 --inflight 1 > $O/write_$wl.log 2>&1
done
# keep the merge small: drop the raw per-dispatch traces except the stats / counter CSVs
find $O -name '*kernel_trace.csv' -size +8M -delete
find $O -name '*.db' -delete
du -sh $O
echo r02 baseline done
