#!/bin/bash
# The priority slot streams (PipelinedRunner.stream_priority, profiles/LOG_r06.md 9a) at every shape, forced RCCL world 1: the
# library's choice ("auto") against an explicit -1 and an explicit 0, and the run without a process group beside them.
#     SHAPES="qws:20:5 normal:20:4 synth4:8:2 synth5:4:1" bash tools/r06/dist_priority_shapes.sh      (workload:steps:warmup)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
P='import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d["value"], d["ms_per_step"], d["config"].get("slot_stream_priority"), d["config"]["launch"][:48])'
for spec in ${SHAPES:-qws:20:5 normal:20:4 synth4:8:2 synth5:4:1}; do
  IFS=: read wl st wu <<< "$spec"
  B="python bench.py --workload $wl --steps $st --warmup $wu --no-cpu-baseline --no-other-precision --no-kernel-timers"
  for i in 1 2; do
    echo -n "$wl dist priority auto: "; GNNPN_FORCE_DIST=1 timeout -k 10 300 $B 2>/dev/null | python3 -c "$P"
    echo -n "$wl dist priority -1:   "; GNNPN_PIPE_STREAM_PRIORITY=-1 GNNPN_FORCE_DIST=1 timeout -k 10 300 $B 2>/dev/null | python3 -c "$P"
    echo -n "$wl dist priority 0:    "; GNNPN_PIPE_STREAM_PRIORITY=0 GNNPN_FORCE_DIST=1 timeout -k 10 300 $B 2>/dev/null | python3 -c "$P"
  done
  echo -n "$wl plain:              "; timeout -k 10 300 $B 2>/dev/null | python3 -c "$P"
done
