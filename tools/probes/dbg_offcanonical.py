"""When do seats go off their canonical CUs in a plain two-slot run?  Phases of 120 steps with a poll between them; per phase the
new off-canonical / declined seats of both slots and the time.  Variants: with / without an eager pass before, bursts of 20."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import WORKLOADS, build_models
import gnnpn_sc_amd.synth as synth
from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
w = dict(WORKLOADS["qws"]); T, K, S, B = w["T"], w["K"], w["S"], w["B"]
dev = torch.device("cuda:0")
table = synth.make_service_table(T, S, seed=0, degree=32)
net, low, high = build_models(T, S, K, dev, w["n_gcn"])
pipe = ML2PNPipeline(net, low, high, K)
svc = DeviceServices.from_table(table, dev)
batch = DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=1, tasks_per_problem=T), dev)
def seats(r):
    for w_ in r.workspaces:
        w_._read()
    return [(w_.last_seats["declined"], w_.last_seats["off_canonical"]) for w_ in r.workspaces]
for variant in ("plain", "eager pass first", "bursts of 20"):
    r = PipelinedRunner(pipe, svc, batch, slots=2, auto_degrade=False)
    if variant == "eager pass first":
        pipe.run(svc, batch, decode_impl=r.decode_impl)
        torch.cuda.synchronize()
    prev = seats(r)
    for phase in range(4):
        t0 = time.perf_counter()
        if variant == "bursts of 20":
            for _ in range(6):
                for _ in range(20):
                    r.submit()
                r.synchronize(check=False)
        else:
            for _ in range(120):
                r.submit()
        word = r.poll()
        dt = time.perf_counter() - t0
        cur = seats(r)
        print(f"{variant}, phase {phase}: {dt * 1e3:.1f} ms, status {word:#x}, new (declined, off-canonical) per slot {[(c[0] - p[0], c[1] - p[1]) for c, p in zip(cur, prev)]}", flush=True)
        prev = cur
    del r
    torch.cuda.synchronize()
