"""How much the matrix pipe gains from MORE resident workgroups per CU: aggregate throughput of the cooperative encoder with
S launches in flight on S streams — the 8-member form (244 registers: two workgroups per CU) at 256 problems per launch
against the 16-member form (168 registers: three per CU) at 128 problems per launch (16 groups of 16: one workgroup per CU
per launch).  QWS shape, two nets per launch as in the pipeline.  python tools/bench_encode_occupancy.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import ops
dev = torch.device("cuda:0")
H, L = 256, 235
g = torch.Generator().manual_seed(0)
def nets_for(B):
    x = torch.rand(B, L, 8, generator=g).to(dev)
    return [{"inputs": x, "w_in": ((torch.rand(4 * H, 8, generator=g) * 2 - 1) * 0.3).to(dev),
             "b_in": ((torch.rand(4 * H, generator=g) * 2 - 1) * 0.3).to(dev),
             "whh": ops.pack_lstm_weight((torch.rand(4 * H, H, generator=g) * 2 - 1) / 16).to(dev),
             "bhh": ((torch.rand(4 * H, generator=g) * 2 - 1) / 16).to(dev)} for _ in range(2)]
def run(impl, B, n_streams, iters=60):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    wss = [ops.new_workspaces(dev) for _ in range(n_streams)]
    args = [nets_for(B) for _ in range(n_streams)]
    for s in range(n_streams):
        with torch.cuda.stream(streams[s]):
            for _ in range(3): ops.lstm_encode(args[s], impl=impl, ws=wss[s])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        for s in range(n_streams):
            with torch.cuda.stream(streams[s]):
                ops.lstm_encode(args[s], impl=impl, ws=wss[s])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for w in wss: w.check("occupancy bench")
    return n_streams * iters * B / dt, dt / iters * 1e3
for impl, B, ns in ((2, 256, 1), (2, 256, 2), (3, 128, 1), (3, 128, 2), (3, 128, 3), (3, 128, 4), (3, 256, 1), (2, 128, 2), (2, 128, 4)):
    try:
        rate, ms = run(impl, B, ns)
        print(f"impl={impl} B={B:4d} launches in flight={ns}: {rate / 1e3:8.1f} k problems/s encoded  ({ms:.3f} ms per round)", flush=True)
    except Exception as e:   # noqa: BLE001
        print(f"impl={impl} B={B} streams={ns}: FAILED {str(e)[:120]}", flush=True)
