"""Solo timings of the recurrent kernels at a bench shape: encoder forms side by side (8-member form, one launch; two co-resident
launches of half the batch each on two streams — what the runner's half-batch mode and two slots do; and, when
tools/experiments/encode_two_tiles_per_workgroup.patch is applied and the library rebuilt, the two-tile form, impl 5).
    python tools/time_recurrent.py [--workload qws] [--batch 512] [--reps 20]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gnnpn_sc_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="qws")
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
w = bench.WORKLOADS[a.workload]
B, L, H = a.batch, w["T"] * w["K"], 256
g = torch.Generator().manual_seed(0)
def net():
    return {"w_in": ((torch.rand(4 * H, 8, generator=g) * 2 - 1) * 0.3).to(dev), "b_in": ((torch.rand(4 * H, generator=g) * 2 - 1) * 0.3).to(dev),
            "whh": ops.pack_lstm_weight((torch.rand(4 * H, H, generator=g) * 2 - 1) / 16).to(dev), "bhh": ((torch.rand(4 * H, generator=g) * 2 - 1) / 16).to(dev)}
x = torch.rand(B, L, 8, generator=g).to(dev)
nets = [dict(net(), inputs=x) for _ in range(2)]
half = B // 2
nets_a = [dict(n, inputs=x[:half].contiguous()) for n in nets]
nets_b = [dict(n, inputs=x[half:].contiguous()) for n in nets]
ws = [ops.new_workspaces(dev) for _ in range(3)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def timed(fn):
    best = 1e9
    for rnd in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            best = min(best, e0.elapsed_time(e1) / a.reps)
    return round(best, 4)

def one(impl):
    return lambda: ops.lstm_encode(nets, precision="split", impl=impl, ws=ws[0])

def two_halves():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        ops.lstm_encode(nets_a, precision="split", impl=2, ws=ws[1], lds_kb=78, paired_start=True)
    with torch.cuda.stream(s2):
        ops.lstm_encode(nets_b, precision="split", impl=2, ws=ws[2], lds_kb=78, paired_start=True)
    cur.wait_stream(s1); cur.wait_stream(s2)

out = {"workload": a.workload, "problems": B, "steps": L, "nets": 2, "precision": "split",
       "ms_8_member_one_launch": timed(one(2)), "ms_two_co_resident_half_batch_launches": timed(two_halves)}
try:
    out["ms_two_tiles_per_workgroup"] = timed(one(5))
    ref = ops.lstm_encode(nets, precision="split", impl=2, ws=ws[0])
    got = ops.lstm_encode(nets, precision="split", impl=5, ws=ws[0])
    out["bit_identical"] = all(torch.equal(r, q) for rr, gg in zip(ref, got) for r, q in zip(rr, gg))
except ops.GnnpnError:
    out["ms_two_tiles_per_workgroup"] = None        # the experiment is not built in
ops.check_status(dev)
print(json.dumps(out))
