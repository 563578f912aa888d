"""Two PipelinedRunners alive in one process and fed alternately (two models served from one process): up to FOUR cooperative launches
want the two workgroup slots of every CU.  Is the result still right, is anything reported, what does it cost?
    python tools/probes/dbg_two_runners.py [--steps 200]"""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import WORKLOADS, build_models
import gnnpn_sc_amd.synth as synth
from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
w = dict(WORKLOADS["qws"])
T, K, S, B = w["T"], w["K"], w["S"], w["B"]
dev = torch.device("cuda:0")
table = synth.make_service_table(T, S, seed=0, degree=32)
runners, refs = [], []
for seed in (0, 1):
    net, low, high = build_models(T, S, K, dev, w["n_gcn"], seed=seed)
    pipe = ML2PNPipeline(net, low, high, K)
    svc = DeviceServices.from_table(table, dev)
    batch = DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=1 + seed, tasks_per_problem=T), dev)
    r = PipelinedRunner(pipe, svc, batch, slots=2, auto_degrade=False)
    runners.append(r)
    refs.append(pipe.run(svc, batch, decode_impl=r.decode_impl))
torch.cuda.synchronize()
for mode in ("one after the other", "alternately"):
    for r in runners:
        r.synchronize(check=True)
    t0 = time.perf_counter()
    if mode == "alternately":
        for i in range(a.steps):
            runners[i % 2].submit()
    else:
        for r in runners:
            for i in range(a.steps // 2):
                r.submit()
            r.synchronize(check=False)
    words = [r.poll() for r in runners]
    dt = time.perf_counter() - t0
    ok = all(torch.equal(r.graphs[s].outputs["idx_high"], ref["idx_high"]) and torch.equal(r.graphs[s].outputs["R"], ref["R"])
             for r, ref in zip(runners, refs) for s in range(2))
    print(f"{mode}: {a.steps} steps in {dt * 1e3:.1f} ms ({B * a.steps / dt / 1e3:.0f} k problems/s), status words {[hex(v) for v in words]}, "
          f"outputs equal to the single-stream runs: {ok}, seats {[w_.last_seats for r in runners for w_ in r.workspaces]}", flush=True)
