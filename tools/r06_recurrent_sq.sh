#!/bin/bash
# Round 6: SQ counters of the two recurrent kernels (lstm_encode_coop_kernel, pointer_decode_lean_kernel), separate --pmc
# passes with the kernel trace only, over the eager single-stream bench pass (`bench.py --graph 0 --inflight 1`):
#     gpurun --timeout 1100 -- 'bash tools/r06_recurrent_sq.sh'       -> gpurun_out/r06sq/summary.json
# then tools/r06_recurrent_sq_summary.py copies it to profiles/r06_recurrent_sq_summary.json.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SOLO="--min-time 0 --no-cpu-baseline --no-kernel-timers --no-other-precision --graph 0 --inflight 1"
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
      "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR")
for cfg in ${CFGS:-qws:split:3 qws:f32:3 synth4:split:1}; do
  wl=${cfg%%:*}; rest=${cfg#*:}; pr=${rest%%:*}; st=${rest#*:}
  i=0
  for set in "${SETS[@]}"; do
    tag=${wl}_${pr}_$i; i=$((i+1))
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/$tag -- python3 $R/bench.py --workload $wl --precision $pr --steps $st --warmup 1 $SOLO > $O/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $O/$tag.log; }
    echo "pass $tag done"
  done
done
find $O -name '*kernel_trace.csv' -size +4M -delete
find $O -name '*.db' -delete
python3 $R/tools/r06_recurrent_sq_summary.py $O
