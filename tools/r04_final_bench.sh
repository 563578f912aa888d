cd $GRAFT_REPO_ROOT; O=gpurun_out/r04prof; mkdir -p $O
timeout 400 python bench.py > $O/bench_qws.json 2> $O/bench_qws.err; echo "qws rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_qws_driver_flags.json 2> /dev/null
timeout 400 python bench.py --workload normal --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_normal.json 2> /dev/null; echo "normal rc=$?"
timeout 500 python bench.py --workload synth4 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_synth4.json 2> $O/bench_synth4.err; echo "synth4 rc=$?"
timeout 500 python bench.py --workload synth5 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_synth5.json 2> /dev/null; echo "synth5 rc=$?"
timeout 200 python tools/bench_pcie.py > $O/pcie_qws.txt 2>&1
timeout 200 python tools/bench_pcie.py --workload synth4 --steps 40 > $O/pcie_synth4.txt 2>&1
