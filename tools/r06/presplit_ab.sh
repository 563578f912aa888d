#!/bin/bash
# A/B on ONE box, alternating fresh processes: the recurrent weights split once per model (default) against split in every launch.
# (GNNPN_PRESPLIT=0 was a temporary switch in modelPN.PointerNet.packed() at commit c6b6052; it is not in the product any more —
#  to repeat the A/B, pop enc_whh_split / dec_whh_split from the packed dicts as test_presplit_weights_change_nothing does.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06presplit
mkdir -p $O
cd $R
for i in 1 2 3; do
for ps in 1 0; do
GNNPN_PRESPLIT=$ps timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision > $O/ab_${ps}_$i.json 2> $O/ab_${ps}_$i.err
python3 -c "
import json
d=json.loads([l for l in open('$O/ab_${ps}_$i.json') if l.startswith('{')][-1])
print('presplit', $ps, d['value'], d['ms_per_step'], [ (k['kernel'],k['avg_ms']) for k in d.get('kernels',[])][:2])
"
done
done
