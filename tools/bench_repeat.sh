#!/bin/bash
# Run-to-run spread of the headline:  bash tools/bench_repeat.sh N [bench args...]
N=$1; shift
for i in $(seq $N); do
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-timers "$@" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['steps'], d['value'], d['ms_per_step'], (d.get('split_operands') or {}).get('value'))"
done
