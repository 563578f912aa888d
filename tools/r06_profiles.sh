#!/bin/bash
# Round-6 measurement artefacts: gpurun --timeout 1200 -- 'bash tools/r06_profiles.sh [part]'.  Outputs: gpurun_out/r06prof/
# part 1 = bench lines, part 2 = rocprofv3 kernel stats, part 3 = PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs), default = all.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06prof
PART=${1:-all}
mkdir -p $O
cd $R
if [ $PART = all ] || [ $PART = 1 ]; then
timeout 400 python bench.py > $O/bench_qws.json 2> $O/bench_qws.err; echo "qws rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_qws_driver_flags.json 2> /dev/null
timeout 400 python bench.py --workload normal --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_normal.json 2> /dev/null; echo "normal rc=$?"
timeout 500 python bench.py --workload synth4 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_synth4.json 2> $O/bench_synth4.err; echo "synth4 rc=$?"
timeout 500 python bench.py --workload synth5 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_synth5.json 2> /dev/null; echo "synth5 rc=$?"
timeout 400 python bench.py --workload synth5 --precision f16 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_synth5_f16.json 2> /dev/null
timeout 300 python bench.py --precision f16 --no-cpu-baseline > $O/bench_qws_f16.json 2> /dev/null
timeout 900 python bench.py --workload synth4 --scaling strong --steps 3 --warmup 1 --batches 2 --no-cpu-baseline --no-other-precision > $O/bench_synth4_strong_g4096_n1.json 2> $O/bench_synth4_strong.err; echo "strong rc=$?"
GNNPN_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision > $O/bench_force_dist_rccl_world1.json 2> $O/bench_force_dist_rccl_world1.err
GNNPN_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 4 --batch 64 --steps 10 --warmup 2 --no-cpu-baseline --no-other-precision --no-kernel-timers > $O/bench_selflaunch_4ranks_shared_gpu.json 2> $O/bench_selflaunch_4ranks_shared_gpu.err; echo "selflaunch rc=$?"
timeout 400 python tools/bench_aggregate.py > $O/aggregate.jsonl 2> /dev/null
timeout 200 python tools/bench_front_half.py > $O/front_half_synth4.json 2> /dev/null
timeout 200 python tools/bench_pcie.py > $O/pcie_qws.txt 2>&1
timeout 200 python tools/bench_pcie.py --workload synth4 --steps 40 > $O/pcie_synth4.txt 2>&1
timeout 300 python tools/bench_slot_parts.py > $O/slot_parts_qws.txt 2>&1
timeout 120 python tools/stamp_decode.py > $O/stamps_decode.txt 2>&1
timeout 100 python tools/stamp_encode.py > $O/stamps_encode.txt 2>&1
timeout 200 python tools/bench_slot_parts.py --workload synth4 --iters 4 > $O/slot_parts_synth4.txt 2>&1
GNNPN_PIPE_LOCKSTEP=0 timeout 100 python tools/slot_overlap.py --rounds 8 > $O/slot_overlap_synth5_free.txt 2>&1
timeout 100 python tools/slot_overlap.py --rounds 6 > $O/slot_overlap_synth5_paired.txt 2>&1
fi
cd /tmp && export TMPDIR=/tmp
SOLO="--min-time 0 --no-cpu-baseline --no-kernel-timers --no-other-precision --graph 0 --inflight 1"
if [ $PART = all ] || [ $PART = 2 ]; then
for wl in ${WLS:-qws normal synth4 synth5}; do
  st=20; [ $wl = normal ] && st=8; [ $wl = synth4 ] && st=3; [ $wl = synth5 ] && st=2
  for pr in split f32; do
    timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${wl}_$pr -- python3 $R/bench.py --workload $wl --precision $pr --steps $st --warmup 2 $SOLO > $O/stats_${wl}_$pr.log 2>&1
  done
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_qws_f16 -- python3 $R/bench.py --precision f16 --steps 20 --warmup 2 $SOLO > $O/stats_qws_f16.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_synth5_f16 -- python3 $R/bench.py --workload synth5 --precision f16 --steps 2 --warmup 1 $SOLO > $O/stats_synth5_f16.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_qws_default -- python3 $R/bench.py --steps 20 --warmup 3 --min-time 0 --no-cpu-baseline --no-kernel-timers --no-other-precision > $O/stats_qws_default.log 2>&1
fi
if [ $PART = all ] || [ $PART = 3 ]; then
for wl in ${WLS:-qws normal synth4 synth5}; do
  for pr in split f32; do
    timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_${wl}_$pr -- python3 $R/bench.py --workload $wl --precision $pr --steps 2 --warmup 1 $SOLO > $O/fetch_${wl}_$pr.log 2>&1
    timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_${wl}_$pr -- python3 $R/bench.py --workload $wl --precision $pr --steps 2 --warmup 1 $SOLO > $O/write_${wl}_$pr.log 2>&1
  done
done
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_qws_f16 -- python3 $R/bench.py --precision f16 --steps 2 --warmup 1 $SOLO > $O/fetch_qws_f16.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_qws_f16 -- python3 $R/bench.py --precision f16 --steps 2 --warmup 1 $SOLO > $O/write_qws_f16.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch_synth5_f16 -- python3 $R/bench.py --workload synth5 --precision f16 --steps 2 --warmup 1 $SOLO > $O/fetch_synth5_f16.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write_synth5_f16 -- python3 $R/bench.py --workload synth5 --precision f16 --steps 2 --warmup 1 $SOLO > $O/write_synth5_f16.log 2>&1
fi
find $O -name '*kernel_trace.csv' -size +4M -delete
find $O -name '*.db' -delete
du -sh $O
echo r06 profiles part $PART done
