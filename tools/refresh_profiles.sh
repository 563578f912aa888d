#!/bin/bash
# Regenerate the round's committed measurement artefacts on the GPU box (run through gpurun from the
# repo root):   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh'
# Outputs land in gpurun_out/refresh/ ; copy the summaries into profiles/ afterwards (tools/collect_profiles.py).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --precision f16 --no-cpu-baseline > $O/bench_f16.json 2> /dev/null
timeout 300 python bench.py --precision split --no-cpu-baseline > $O/bench_split.json 2> /dev/null
timeout 200 python tools/split_encode.py > $O/split_accuracy.txt 2> /dev/null
timeout 300 python bench.py --workload normal --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_normal.json 2> /dev/null
timeout 300 python bench.py --workload synth4 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_synth4.json 2> /dev/null
timeout 200 python tools/bench_aggregate.py > $O/aggregate_roofline.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-split-line > $O/prof_default.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_solo -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-split-line --graph 0 --inflight 1 > $O/prof_solo.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-split-line --graph 0 --inflight 1 > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-split-line --graph 0 --inflight 1 > $O/pmc_write.log 2>&1
echo refresh done
