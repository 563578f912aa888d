"""Eager operator calls against HIP-graph replay of the SAME launches, one stream, at the QWS shape (batch 256): what the host side of
the custom-op layer costs per step.  `two_level_greedy` on resident candidate rows (encoder + decoder + reward: the recurrent half
of the path), and the whole pass (`ML2PNPipeline.run`).
    python tools/bench_eager.py [--workload qws] [--steps 200]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_models
import gnnpn_sc_amd.synth as synth
from gnnpn_sc_amd import ops
from gnnpn_sc_amd.modelPN import two_level_greedy
from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="qws")
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
w = dict(WORKLOADS[a.workload])
T, K, S, B = w["T"], w["K"], w["S"], w["B"]
dev = torch.device("cuda:0")
table = synth.make_service_table(T, S, seed=0, degree=32)
net, low, high = build_models(T, S, K, dev, w["n_gcn"])
pipe = ML2PNPipeline(net, low, high, K)
svc = DeviceServices.from_table(table, dev)
batch = DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=1, tasks_per_problem=w["n_t"]), dev)
rows, ids = pipe.candidates(svc, batch, pipe.scores(svc, batch))


def timed(fn, steps):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    return best * 1e3


def host_only(fn, steps):
    """Host time per call with the GPU far behind never the limit: time to ENQUEUE (no synchronise inside the loop)."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    t = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    return t * 1e3


def graph_of(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


out = {"workload": a.workload, "problems": B, "precision": pipe.precision}
for name, fn in (("two_level_greedy", lambda: two_level_greedy(low, high, rows, precision=pipe.precision)),
                 ("whole_pass", lambda: pipe.run(svc, batch))):
    eager = timed(fn, a.steps)
    enqueue = host_only(fn, 50)
    g, _ = graph_of(fn)
    replay = timed(g.replay, a.steps)
    out[name] = {"eager_ms": round(eager, 4), "graph_replay_ms": round(replay, 4), "eager_over_replay": round(eager / replay, 4),
                 "host_enqueue_ms_per_call": round(enqueue, 4)}
ops.check_status()
print(json.dumps(out))
