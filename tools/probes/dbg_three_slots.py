"""PipelinedRunner with more than two slots: three or four replays in flight want the two workgroup slots of every CU."""
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
for slots in (1, 2, 3, 4):
    r = PipelinedRunner(pipe, svc, batch, slots=slots, auto_degrade=False)
    ref = pipe.run(svc, batch, decode_impl=r.decode_impl)
    for _ in range(8):
        r.submit()
    r.synchronize(check=False); r.poll()
    t0 = time.perf_counter()
    for _ in range(120):
        r.submit()
    word = r.poll()
    dt = time.perf_counter() - t0
    ok = all(torch.equal(r.graphs[s].outputs["idx_high"], ref["idx_high"]) for s in range(slots))
    print(f"slots {slots}: streams {r.n_streams}, 120 steps {dt * 1e3:.1f} ms ({B * 120 / dt / 1e3:.0f} k problems/s), status {word:#x}, equal {ok}, "
          f"seats {[w_.last_seats for w_ in r.workspaces]}", flush=True)
    del r
    torch.cuda.synchronize()
print("allocated MB", torch.cuda.memory_allocated() / 1e6)
