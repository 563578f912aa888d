#!/bin/bash
# Soaks of the final library of round 6 (explicit-store hand-off, priority slot streams beside RCCL): outputs gpurun_out/r06soak/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06soak
mkdir -p $O
cd $R
SOAK_DIST=1 SOAK_POLL_EVERY=500 timeout -k 10 420 python tools/soak_pipeline.py 20000 split > $O/split_rccl.txt 2>&1; echo "split+rccl rc=$?"; tail -1 $O/split_rccl.txt | cut -c1-200
SOAK_POLL_EVERY=100 timeout -k 10 420 python tools/soak_pipeline.py 20000 f32 > $O/f32.txt 2>&1; echo "f32 rc=$?"; tail -1 $O/f32.txt | cut -c1-200
SOAK_POLL_EVERY=50 timeout -k 10 300 python tools/soak_pipeline.py 300 split synth4 > $O/synth4.txt 2>&1; echo "synth4 rc=$?"; tail -1 $O/synth4.txt | cut -c1-200
SOAK_DIST=1 SOAK_POLL_EVERY=50 timeout -k 10 300 python tools/soak_pipeline.py 300 split synth4 > $O/synth4_rccl.txt 2>&1; echo "synth4+rccl rc=$?"; tail -1 $O/synth4_rccl.txt | cut -c1-200
