#!/bin/bash
# A/B an environment switch over repeated bench runs:  bash tools/ab_env.sh VAR "0 1 0 1" [bench args...]
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v timeout 300 python bench.py --no-cpu-baseline --no-kernel-timers "$@" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], d['ms_per_step'], (d.get('split_operands') or {}).get('value'))"
done
