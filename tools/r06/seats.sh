#!/bin/bash
# Round 6: the seat protocol's independent accesses in one round trip — placement tests, fixed part, headline, soak
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06seats
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests/test_gpu_pn.py tests/test_gpu_pipeline.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/time_fixed_part.py > $O/fixed_part.jsonl 2> $O/fixed_part.err; echo "fixed rc=$?"; cat $O/fixed_part.jsonl
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads([l for l in open('$O/bench_$i.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], [ (k['kernel'],k['avg_ms']) for k in d.get('kernels',[])][:2], d['per_rank'][0].get('seats'), d['per_rank'][0].get('placement_last_launch'))
"
done
