"""Two cooperative encoder launches in flight on two streams vs one stream (timing experiment)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import ops
dev = torch.device("cuda:0")
B, L, H = 256, 235, 256
g = torch.Generator().manual_seed(0)
x = torch.rand(B, L, 8, generator=g).to(dev)
nets = []
for n in range(2):
    nets.append({"inputs": x, "w_in": ((torch.rand(4 * H, 8, generator=g) * 2 - 1) * 0.3).to(dev),
                 "b_in": ((torch.rand(4 * H, generator=g) * 2 - 1) * 0.3).to(dev),
                 "whh": ops.pack_lstm_weight((torch.rand(4 * H, H, generator=g) * 2 - 1) / 16).to(dev),
                 "bhh": ((torch.rand(4 * H, generator=g) * 2 - 1) / 16).to(dev)})
def run(n_streams, reps=20):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    for i in range(4):
        ops.set_workspace_slot(i % n_streams)
        with torch.cuda.stream(streams[i % n_streams]): ops.lstm_encode(nets)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for i in range(reps):
        ops.set_workspace_slot(i % n_streams)
        with torch.cuda.stream(streams[i % n_streams]): ops.lstm_encode(nets)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    ops.set_workspace_slot(0)
    print(f"{n_streams} stream(s): {ms:.3f} ms per encode launch (2 nets x 256 problems)")
run(1); run(2); run(3)
ops.check_status(dev)

# phase stamps while two launches share the CUs
names = ["sweep+LDS fill", "barrier 1", "A-frag reads + MFMA (+flush)", "projection + barrier 2", "cell + publish"]
ops.set_option("lstm_ablate", 32)
streams = [torch.cuda.Stream() for _ in range(2)]
for rep in range(3):
    for i in range(2):
        ops.set_workspace_slot(i)
        with torch.cuda.stream(streams[i]): ops.lstm_encode(nets)
torch.cuda.synchronize()
for i in range(2):
    ops.set_workspace_slot(i)
    ws = ops.encode_workspace(dev)
    prof = ws[32:32 + 48].view(torch.int64).cpu().tolist()
    n = max(prof[5], 1)
    print(f"slot {i} (co-running):", {k: round(v / n) for k, v in zip(names, prof[:5])}, "total", round(sum(prof[:5]) / n))
ops.set_workspace_slot(0)
ops.set_option("lstm_ablate", 0)
