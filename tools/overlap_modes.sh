#!/bin/bash
# Kernel traces of several default bench runs (the two-slot pipeline locks into a faster or a slower phase
# alignment): gpurun_out/modes/run$i/{trace csv, value.txt}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/modes
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4; do
  rocprofv3 --kernel-trace --output-format csv -d $O/run$i -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-timers --no-split-line > $O/run$i.log 2>&1
  tail -1 $O/run$i.log | cut -c1-200 > $O/run$i.value.txt
done
echo done
