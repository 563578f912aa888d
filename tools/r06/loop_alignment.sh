#!/bin/bash
# Is the encoder's step time sensitive to WHERE its step loop lies in the instruction stream?  Trees _aN: this tree built with
# -DGNNPN_ENC_PAD_NOPS=N (N s_nop in front of the loop); _b5: the tree before the staffing-count fix (same loop, 6 instructions earlier).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
for t in ${TREES:-_a0 _a4 _a8 _a12 _b5 _a0 _a4 _a8 _a12 _b5}; do cd $R/$t; echo "== $t"; timeout -k 10 120 python $R/tools/probes/dbg_solo_placement.py 2>&1 | grep "^lds_kb 0" | cut -c1-110; done
