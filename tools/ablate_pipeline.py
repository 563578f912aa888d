"""Two-slot pipeline throughput under the encoder's diagnostic ablations (results are WRONG under ablation; timing only):
which part of the encoder step the pipelined mode is sensitive to.  python tools/ablate_pipeline.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import ops, synth
from gnnpn_sc_amd.pipeline import ML2PNPipeline, DeviceServices, DeviceBatch, PipelinedRunner
import bench
dev = torch.device("cuda:0")
w = bench.WORKLOADS["qws"]
table = synth.make_service_table(w["T"], w["S"], seed=0, degree=32)
net, low, high = bench.build_models(w["T"], w["S"], w["K"], dev, w["n_gcn"])
svc = DeviceServices.from_table(table, dev)
batch = DeviceBatch.from_problems(synth.make_problem_batch(table, w["B"], seed=1, tasks_per_problem=w["n_t"]), dev)
for label, abl in (("production build", 0), ("no MFMA", 1), ("no transcendentals", 2),
                   ("no tag wait", 4), ("no sweep", 8), ("no sweep, no publish", 24), ("no enc_out flush, no input prefetch", 0x600)):
    ops.set_option("lstm_ablate", abl)
    pipe = ML2PNPipeline(net, low, high, w["K"])
    runner = PipelinedRunner(pipe, svc, batch, slots=2)
    for _ in range(20):
        runner.submit()
    runner.synchronize(check=False)
    t0 = time.perf_counter()
    n = 400
    for _ in range(n):
        runner.submit()
    runner.synchronize(check=False)
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{label:40s} ablate={abl:5d}  {ms:.4f} ms/step  {w['B'] / ms:.1f} k problems/s", flush=True)
    for x in runner.workspaces:
        x.status.zero_()
    del runner, pipe
ops.set_option("lstm_ablate", 0)
