#!/bin/bash
# Round-2 measurement artefacts: gpurun --timeout 2400 -- 'bash tools/r02_profiles.sh'.  Outputs: gpurun_out/r02prof/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02prof
mkdir -p $O
cd $R
timeout 300 python bench.py > $O/bench_qws.json 2> $O/bench_qws.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_qws_driver_flags.json 2> /dev/null
timeout 400 python bench.py --workload normal --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_normal.json 2> /dev/null
timeout 400 python bench.py --workload synth4 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_synth4.json 2> /dev/null
timeout 900 python bench.py --workload synth4 --scaling strong --steps 3 --warmup 1 --batches 2 --no-cpu-baseline --no-split-line > $O/bench_synth4_strong_g4096_n1.json 2> $O/bench_synth4_strong.err
timeout 400 python bench.py --workload synth5 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_synth5.json 2> /dev/null
timeout 400 python bench.py --workload synth5 --precision f16 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_synth5_f16.json 2> /dev/null
timeout 300 python bench.py --precision f16 --no-cpu-baseline > $O/bench_qws_f16.json 2> /dev/null
GNNPN_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-split-line > $O/bench_force_dist_rccl_world1.json 2> $O/bench_force_dist_rccl_world1.err
GNNPN_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline --no-split-line --no-kernel-timers > $O/bench_selflaunch_2ranks_shared_gpu.json 2> $O/bench_selflaunch_2ranks_shared_gpu.err; echo "selflaunch rc=$?"
GNNPN_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --scaling strong --global-batch 256 --steps 10 --warmup 2 --no-cpu-baseline --no-split-line --no-kernel-timers > $O/bench_selflaunch_2ranks_strong.json 2> $O/bench_selflaunch_2ranks_strong.err; echo "selflaunch strong rc=$?"
timeout 300 python tools/bench_aggregate.py > $O/aggregate.jsonl 2> /dev/null
cd /tmp && export TMPDIR=/tmp
for wl in qws normal synth4; do
  st=20; [ $wl = normal ] && st=8; [ $wl = synth4 ] && st=3
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python3 $R/bench.py --workload $wl --steps $st --warmup 2 --min-time 0 --no-cpu-baseline --no-kernel-timers --no-split-line --graph 0 --inflight 1 > $O/stats_$wl.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_$wl -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --min-time 0 --no-cpu-baseline --no-kernel-timers --no-split-line --graph 0 --inflight 1 > $O/fetch_$wl.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_$wl -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --min-time 0 --no-cpu-baseline --no-kernel-timers --no-split-line --graph 0 --inflight 1 > $O/write_$wl.log 2>&1
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_qws_default -- python3 $R/bench.py --steps 20 --warmup 3 --min-time 0 --no-cpu-baseline --no-kernel-timers --no-split-line > $O/stats_qws_default.log 2>&1
find $O -name '*kernel_trace.csv' -size +4M -delete
find $O -name '*.db' -delete
du -sh $O
echo r02 profiles done
