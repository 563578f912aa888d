#!/bin/bash
# alternating runs, fresh process each: gate off; front kernels padded to 78 KB or not
O=gpurun_out/r05_slip; mkdir -p $O
for i in 1 2 3; do
  for cfg in "0:0" "0:78" "400:0"; do
    us=${cfg%%:*}; kb=${cfg#*:}
    GNNPN_PIPE_COMMON_START_US=$us GNNPN_PIPE_FRONT_LDS_KB=$kb timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-precision --no-kernel-timers > $O/run_${us}_${kb}_$i.json 2> $O/run_${us}_${kb}_$i.err
    python - <<PY
import json
d=json.loads([l for l in open("$O/run_${us}_${kb}_$i.json") if l.startswith("{")][-1])
r=d["timing"]["round_us"]; m=sorted(r)[len(r)//2]
slow=[v for v in r if v>1.025*m]
print(json.dumps({"common_start_us":$us,"front_lds_kb":$kb,"run":$i,"value":round(d["value"]),"median_round_us":m,"rounds":len(r),"slow_rounds_gt_2.5pct":len(slow),"slow_excess_us":[v-m for v in slow][:12],"declined":[p["declined_seats"] for p in d["per_rank"][0]["placement_last_launch"]]}))
PY
  done
done
