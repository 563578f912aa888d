#!/bin/bash
# gpurun --timeout 1800 -- 'bash tools/r02_gpu_check.sh [tag]': the -m gpu suite, the default bench line, and the RCCL path
# exercised with one rank (GNNPN_FORCE_DIST=1).  Outputs: gpurun_out/<tag>/ and gpurun_out/parity/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02check}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
GNNPN_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-line > $O/bench_force_dist_rccl_world1.json 2> $O/bench_force_dist_rccl_world1.err; echo "force_dist rc=$?"
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
head -c 600 $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_solo -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-split-line --graph 0 --inflight 1 --min-time 0 > $O/prof_solo.log 2>&1
f=$(ls -t $O/prof_solo/*/*kernel_stats.csv | head -1); head -25 $f
