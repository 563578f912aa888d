#!/bin/bash
# Round-4 closing measurements that r04_profiles.sh does not cover: gpurun --timeout 1200 -- 'bash tools/r04_final.sh'
# (the ablation libraries are built beforehand, here in the container: python tools/ablate_aggregate.py build; python tools/ablate_gin_layer.py build)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04final
mkdir -p $O
cd $R
{ rocm-smi --showbus --showserial --showuniqueid 2>/dev/null | grep -v "^=\|^$"; python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print({"name": p.name, "cus": p.multi_processor_count, "arch": getattr(p, "gcnArchName", "?"), "mem_GB": round(p.total_memory / 2**30)})
PY
} > $O/box.txt 2>&1
timeout -k 10 150 python tools/probes/dbg_soak.py 16 "" > $O/stress.jsonl 2>&1; echo "stress rc=$?"
timeout -k 10 200 python tools/bench_eager.py > $O/eager_vs_graph_qws.json 2> /dev/null; echo "eager rc=$?"
timeout -k 10 120 python tools/bench_gin_layer.py > $O/gin_layer_bench.json 2> /dev/null; echo "gin rc=$?"
timeout -k 10 150 python tools/bench_front_half.py > $O/front_half_synth4.json 2> /dev/null
timeout -k 10 400 python tools/ablate_gin_layer.py run > $O/gin_layer_ablation.jsonl 2>&1; echo "gin ablation rc=$?"
timeout -k 10 500 python tools/ablate_aggregate.py run > $O/csr_aggregate_ablation.jsonl 2>&1; echo "aggregate ablation rc=$?"
bash tools/r04_gin_pmc.sh > $O/gin_pmc.log 2>&1; cp $R/gpurun_out/r04ginpmc/summary.json $O/gin_layer_pmc_summary.json 2>/dev/null
du -sh $O
echo r04 final done
