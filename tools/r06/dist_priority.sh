#!/bin/bash
# Slot streams of a priority of their own (pipeline.py, PipelinedRunner.stream_priority) against more hardware queues, as a way of
# keeping the collective's stream wait out of the slots' hardware queues.  `grid` = the four combinations, value only; `full` = the
# library's default (auto priority, 4 queues) against round 6's earlier default (8 queues, normal priority) with the whole line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
P='import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d["value"], d["ms_per_step"], d["config"].get("slot_stream_priority"), (d.get("pcie_inclusive") or {}).get("ratio_to_value"), [r.get("collective_ms") for r in d.get("per_rank", [])][:1])'
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision"
if [ "${1:-grid}" = grid ]; then
for i in 1 2; do
for cfg in "4 0" "4 -1" "8 0" "8 -1"; do
  set -- $cfg
  echo -n "queues $1 priority $2 dist: "; GPU_MAX_HW_QUEUES=$1 GNNPN_PIPE_STREAM_PRIORITY=$2 GNNPN_FORCE_DIST=1 timeout -k 10 300 $B 2>/dev/null | python3 -c "$P"
done
echo -n "queues 4 priority -1 plain: "; GPU_MAX_HW_QUEUES=4 GNNPN_PIPE_STREAM_PRIORITY=-1 timeout -k 10 300 $B 2>/dev/null | python3 -c "$P"
echo -n "queues 4 priority 0 plain: "; GPU_MAX_HW_QUEUES=4 timeout -k 10 300 $B 2>/dev/null | python3 -c "$P"
done
else
for i in 1 2 3; do
  echo -n "dist, library default: "; GNNPN_FORCE_DIST=1 timeout -k 10 300 $B 2>/dev/null | tee gpurun_out/prio_auto_$i.json | python3 -c "$P"
  echo -n "dist, 8 queues priority 0: "; GPU_MAX_HW_QUEUES=8 GNNPN_PIPE_STREAM_PRIORITY=0 GNNPN_FORCE_DIST=1 timeout -k 10 300 $B 2>/dev/null | tee gpurun_out/prio_q8_$i.json | python3 -c "$P"
  echo -n "plain, library default: "; timeout -k 10 300 $B 2>/dev/null | tee gpurun_out/prio_plain_$i.json | python3 -c "$P"
done
fi
