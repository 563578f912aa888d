#!/bin/bash
# rocprofv3 kernel stats of the ES-WOA fine-tuner at the reference's parameters -> gpurun_out/woa/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/woa
rm -rf $O; mkdir -p $O
cd $R && timeout 300 python tests/campaigns/bench_woa.py 1000 500 > $O/bench_woa.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tests/campaigns/bench_woa.py 1000 500 > $O/prof.log 2>&1
cat $O/bench_woa.txt
