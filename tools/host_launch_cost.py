"""Host-side cost of one PipelinedRunner.submit() (a hipGraphLaunch of the whole pass) vs the GPU time per step."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import synth
from gnnpn_sc_amd.pipeline import ML2PNPipeline, DeviceServices, DeviceBatch, PipelinedRunner
import bench
dev = torch.device("cuda:0")
w = bench.WORKLOADS["qws"]
table = synth.make_service_table(w["T"], w["S"], seed=0, degree=32)
pb = synth.make_problem_batch(table, w["B"], seed=1, tasks_per_problem=w["n_t"])
net, low, high = bench.build_models(w["T"], w["S"], w["K"], dev, w["n_gcn"])
pipe = ML2PNPipeline(net, low, high, w["K"])
svc, batch = DeviceServices.from_table(table, dev), DeviceBatch.from_problems(pb, dev)
runner = PipelinedRunner(pipe, svc, batch, slots=2)
for _ in range(20): runner.submit()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n): runner.submit()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host: {1e3 * (t1 - t0) / n:.3f} ms per submit; whole: {1e3 * (t2 - t0) / n:.3f} ms per step")
