#!/bin/bash
# Soaks with a random spin kernel (0..600 us) in front of half of the submits: the slots drift through every phase relation
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06soak
mkdir -p $O
cd $R
SOAK_JITTER_US=600 SOAK_POLL_EVERY=250 timeout -k 10 500 python tools/soak_pipeline.py 10000 split > $O/jitter_split.txt 2>&1; echo "split rc=$?"; tail -1 $O/jitter_split.txt | cut -c1-260
SOAK_JITTER_US=600 SOAK_POLL_EVERY=250 timeout -k 10 500 python tools/soak_pipeline.py 6000 f32 > $O/jitter_f32.txt 2>&1; echo "f32 rc=$?"; tail -1 $O/jitter_f32.txt | cut -c1-260
SOAK_JITTER_US=600 SOAK_DIST=1 SOAK_POLL_EVERY=250 timeout -k 10 500 python tools/soak_pipeline.py 6000 split > $O/jitter_split_rccl.txt 2>&1; echo "split+rccl rc=$?"; tail -1 $O/jitter_split_rccl.txt | cut -c1-260
SOAK_JITTER_US=20000 SOAK_POLL_EVERY=20 timeout -k 10 500 python tools/soak_pipeline.py 100 split synth5 > $O/jitter_synth5.txt 2>&1; echo "synth5 rc=$?"; tail -1 $O/jitter_synth5.txt | cut -c1-260
SOAK_JITTER_US=3000 SOAK_POLL_EVERY=100 timeout -k 10 500 python tools/soak_pipeline.py 1000 split normal > $O/jitter_normal.txt 2>&1; echo "normal rc=$?"; tail -1 $O/jitter_normal.txt | cut -c1-260
