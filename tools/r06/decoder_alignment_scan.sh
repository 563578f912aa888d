#!/bin/bash
# The decoder's step loop moved through a 64-byte line in 4-byte steps (trees _dN: decode_lean.hip with `asm volatile(".p2align 6; .rept N; s_nop 0; .endr")`
# in front of `for (int k = 0; k <= T; ++k)`, built with -DGNNPN_DEC_PAD_NOPS=N — the hook is not in the product source: the scan came out flat):
# the driver's command per tree — headline, and the solo (eager, one stream) decoder and encoder milliseconds of the same run.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for n in ${NS:-0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15}; do
  cd $R/_d$n
  echo -n "nops $n: "; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k={x['kernel']:x['avg_ms'] for x in d['kernels']}
print('headline', d['value'], 'solo decoder ms', k['pointer_decode'], 'solo encoder ms', k['lstm_encode'])"
done
done
