"""Where the two-slot pipeline gains or loses: encoder launches alone, decoder launches alone (the 2-per-CU build) and whole
two-level passes, each with 1 and 2 launches in flight on separate streams, per precision.  QWS shape B=256 by default.
    python tools/bench_slot_parts.py [--workload qws|normal|synth4] [--batch B]"""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_models
from gnnpn_sc_amd import custom_ops, ops
from gnnpn_sc_amd.modelPN import two_level_greedy

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="qws")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--iters", type=int, default=40)
a = ap.parse_args()
w = WORKLOADS[a.workload]
T, K, B = w["T"], w["K"], a.batch or w["B"]
dev = torch.device("cuda:0")
_, low, high = build_models(T, w["S"], K, dev)
g = torch.Generator().manual_seed(0)


def inputs():
    x = torch.rand(B, T * K, 8, generator=g)
    x[:, :, 2:4] = 0.9 + 0.1 * x[:, :, 2:4]
    x[:, K:, 4:] = 0
    return x.to(dev)


def timed(fns, iters):
    """fns: one callable per stream; -> ms per round of len(fns) launches."""
    streams = [torch.cuda.Stream() for _ in fns]
    for s, f in zip(streams, fns):
        with torch.cuda.stream(s):
            for _ in range(3):
                f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for s, f in zip(streams, fns):
            with torch.cuda.stream(s):
                f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for prec in ("f32", "split"):
    la, ha = low.actor, high.actor
    for n in (1, 2):
        lds = 78 if (prec == "split" and n > 1) else 0      # one LDS footprint for co-resident launches (pipeline.py)
        xs = [inputs() for _ in range(n)]
        wss = [ops.new_workspaces(dev) for _ in range(n)]
        encs = []
        def enc_fn(i):
            el, eh = la.encode_args(xs[i])[0], ha.encode_args(xs[i])[0]
            return lambda: custom_ops.lstm_encode([el, eh], precision=prec, lds_kb=lds, ws=wss[i])
        ms_e = timed([enc_fn(i) for i in range(n)], a.iters)
        outs = [enc_fn(i)() for i in range(n)]
        torch.cuda.synchronize()
        def dec_fn(i, impl):
            enc, h_n, c_n = outs[i]
            args = [la.decode_args(None, enc[0], h_n[0], c_n[0]), ha.decode_args(None, enc[1], h_n[1], c_n[1], latent_from=0)]
            return lambda: custom_ops.pointer_decode(args, xs[i], T, K, la.C, la.use_tanh, precision=prec, impl=impl, lds_kb=lds, ws=wss[i])
        ms_d4 = timed([dec_fn(i, 4) for i in range(n)], a.iters)
        ms_d2 = timed([dec_fn(i, 2) for i in range(n)], a.iters) if n == 1 else float("nan")
        ms_t = timed([(lambda i=i: two_level_greedy(low, high, xs[i], precision=prec, decode_impl=4, lds_kb=lds, ws=wss[i])) for i in range(n)], a.iters)
        for ws_ in wss:
            ws_.check("bench_slot_parts")
        print(f"{prec:5s} {n} in flight: encoder {ms_e:.3f} ms/round ({n * B / ms_e:.0f} k problems/s)   decoder(2-per-CU build) {ms_d4:.3f}"
              f"   decoder(1-per-CU build) {ms_d2:.3f}   two-level pass {ms_t:.3f} ms/round ({n * B / ms_t:.0f} k problems/s)", flush=True)
