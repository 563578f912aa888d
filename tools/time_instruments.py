"""Which clock does a kernel duration come from?  (rocprofv3's per-kernel durations of the tiled aggregate are
7-13 % longer than tools/bench_aggregate.py's back-to-back HIP events.)  One launch shape, three instruments in ONE process:

  b2b     : e0, N launches back to back, e1 — (e1 - e0) / N, per round (what bench_aggregate.py::timed takes the minimum of)
  single  : synchronize, e0, ONE launch, e1, synchronize — the duration of one launch with nothing queued behind or before it
  paired  : e0, launch, e1 recorded around EVERY launch of a back-to-back train — per-launch begin-to-end inside the train

Run it plain and under `rocprofv3 --kernel-trace --stats` (same process, same launches): the profiler's per-dispatch durations of the
same launches are then comparable with all three, and the plain run says whether the profiler itself moves the numbers.

    python tools/time_instruments.py [--config 5000:128] [--reps 20] [--rounds 6] [--form default]
"""
import argparse, json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import graph, ops, synth

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="5000:128")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--form", default="default", help="default | 0 (one tile buffer) | 1 (two tile buffers)")
a = ap.parse_args()
dev = torch.device("cuda:0")
S, copies = (int(v) for v in a.config.split(":"))
table = synth.make_service_table(47, S, 0, degree=32, graph="scan")
csr = graph.gcn_csr(torch.from_numpy(table.edge_index), torch.from_numpy(table.edge_attr), S)
nnz = csr.col.numel()
rp = torch.cat([csr.rowptr[:-1].long() + c * nnz for c in range(copies)] + [torch.tensor([copies * nnz])]).int().to(dev)
col = torch.cat([csr.col.long() + c * S for c in range(copies)]).int().to(dev)
w = csr.w.repeat(copies).to(dev)
N, C = copies * S, 256
norm = ops.gcn_norm(rp, col, w)
g = torch.Generator(device=dev).manual_seed(S + copies)
x = torch.randn(N, C, device=dev, generator=g)
bias = torch.randn(C, device=dev, generator=g)
scale, shift = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g)
plan = ops.TilePlan(rp, col, norm, S) if a.form == "default" else ops.TilePlan(rp, col, norm, S, form=int(a.form))
assert plan.valid
fn = lambda: plan.aggregate(x, None, bias, scale, shift, ops.ACT_RELU)   # noqa: E731
alg = 2 * N * C * 4 + nnz * copies * 8 + (N + 1) * 4
for _ in range(3 * a.reps):
    fn()
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
b2b, single, paired = [], [], []
for _ in range(a.rounds):
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    b2b.append(e0.elapsed_time(e1) / a.reps)
for _ in range(a.rounds * a.reps // 2):
    e0, e1 = ev(), ev()
    torch.cuda.synchronize()
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    single.append(e0.elapsed_time(e1))
for _ in range(a.rounds):
    es = [(ev(), ev()) for _ in range(a.reps)]
    for s0, s1 in es:
        s0.record()
        fn()
        s1.record()
    torch.cuda.synchronize()
    paired.append([s0.elapsed_time(s1) for s0, s1 in es])
flat = [v for r in paired for v in r[1:]]
q = lambda v, p: sorted(v)[min(len(v) - 1, int(p * len(v)))]   # noqa: E731
rec = {"what": "tiled aggregate, one shape, three HIP-event instruments in one process", "config": a.config, "form": plan.geom.get("form", 0),
       "algorithmic_bytes": alg, "under_rocprof": bool(os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD") or os.environ.get("ROCP_TOOL_LIBRARIES")),
       "b2b_ms": {"rounds": [round(v, 4) for v in b2b], "min": round(min(b2b), 4), "median": round(statistics.median(b2b), 4)},
       "single_ms": {"n": len(single), "min": round(min(single), 4), "median": round(statistics.median(single), 4), "p90": round(q(single, 0.9), 4),
                     "mean": round(statistics.mean(single), 4)},
       "paired_ms": {"n": len(flat), "min": round(min(flat), 4), "median": round(statistics.median(flat), 4), "mean": round(statistics.mean(flat), 4),
                     "first_of_train_mean": round(statistics.mean(r[0] for r in paired), 4)}}
for k in ("b2b_ms", "single_ms", "paired_ms"):
    rec[k]["frac_of_8TBps_at_median"] = round(alg / rec[k]["median"] / 8e9, 4)
print(json.dumps(rec), flush=True)
