"""Experiment: the two slots forced to start every step together (in phase) vs free running."""
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
for prec in os.environ.get("PRECS", "f32,split").split(","):
    pipe = ML2PNPipeline(net, low, high, w["K"], precision=prec)
    svc, batch = DeviceServices.from_table(table, dev), DeviceBatch.from_problems(pb, dev)
    runner = PipelinedRunner(pipe, svc, batch, slots=2)
    n = 100
    for mode in os.environ.get("MODES", "free,lockstep,free").split(","):
        for _ in range(10): runner.submit()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(), torch.cuda.Event()]
        t0 = time.perf_counter()
        for i in range(n):
            if mode == "free":
                runner.submit(); runner.submit()
            else:
                runner.submit(); ev[0].record(runner.stream(0))
                runner.submit(); ev[1].record(runner.stream(1))
                runner.stream(0).wait_event(ev[1]); runner.stream(1).wait_event(ev[0])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(prec, mode, f"{1e3 * dt / (2 * n):.4f} ms/step  {w['B'] * 2 * n / dt:.0f} problems/s")
