#!/bin/bash
# Repro loop for the two-slot synth4 bench (diagnosis): N runs, each prints its JSON line or the failure record.
N=${1:-4}; shift
mkdir -p gpurun_out/repro
for i in $(seq 1 $N); do
  timeout 300 python bench.py --workload synth4 --steps 8 --warmup 2 --no-cpu-baseline --no-split-line --no-kernel-timers "$@" \
    > gpurun_out/repro/run$i.out 2> gpurun_out/repro/run$i.err
  echo "run $i rc=$?"; tail -c 600 gpurun_out/repro/run$i.out; grep -o "status 0x.*" gpurun_out/repro/run$i.err | tail -c 3000
done
