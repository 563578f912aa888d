#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/badmode
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/run -- python3 $R/tools/slot_phase.py > $O/run.log 2>&1
grep "quarter" $O/run.log
