"""Per-launch duration of the cooperative encoder, launch after launch on one stream (solo): looks for slow launches
(placement reserve time-outs show as +2 ms and a non-zero off-canonical seat count in the status area).
    python tools/encode_launch_times.py [--n 300] [--first f32|split]"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_models
from gnnpn_sc_amd import custom_ops, ops
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=300)
ap.add_argument("--order", default="f32,split,f32,split")
ap.add_argument("--lds", type=int, default=0)
a = ap.parse_args()
w = WORKLOADS["qws"]
T, K, B = w["T"], w["K"], w["B"]
dev = torch.device("cuda:0")
_, low, high = build_models(T, w["S"], K, dev)
g = torch.Generator().manual_seed(0)
x = torch.rand(B, T * K, 8, generator=g).to(dev)
ws = ops.new_workspaces(dev)
el, eh = low.actor.encode_args(x)[0], high.actor.encode_args(x)[0]
for prec in a.order.split(","):
    ts, stats = [], []
    for i in range(a.n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        custom_ops.lstm_encode([el, eh], precision=prec, lds_kb=a.lds, ws=ws)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
        stats.append(ws.encode()[:16].view(torch.int32).tolist())
    t = torch.tensor(ts)
    slow = [(i, round(ts[i], 3), stats[i]) for i in range(a.n) if ts[i] > 1.5 * float(t.median())]
    print(f"{prec}: median {float(t.median()):.3f} ms, min {float(t.min()):.3f}, max {float(t.max()):.3f}, {len(slow)} of {a.n} launches slower than 1.5 x median", flush=True)
    for s in slow[:8]:
        print("    launch", s[0], "ms", s[1], "status area [status, same-XCD workgroups, off-canonical seats, -]", s[2], flush=True)
