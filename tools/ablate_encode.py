"""Timing experiments on the cooperative encoder (results are WRONG under ablation; timing only)."""
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
def timeit(label, abl):
    ops.set_option("lstm_ablate", abl)
    for _ in range(3): ops.lstm_encode(nets)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.lstm_encode(nets)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{label:40s} ablate={abl:2d}  {ms:.3f} ms  {ms * 1e3 / L:.2f} us/step", flush=True)
timeit("full", 0)
timeit("diagnostic build, stamps on", 32)
timeit("full", 0)
timeit("diagnostic build, stamps on", 32)
timeit("no enc_out flush", 0x200)
timeit("no input prefetch", 0x400)
timeit("neither", 0x600)
timeit("no MFMA", 1)
timeit("no transcendentals", 2)
timeit("no MFMA, no transcendentals", 3)
timeit("no tag wait (one pass)", 4)
timeit("no sweep at all", 8)
timeit("no sweep, no publish", 24)
timeit("no sweep/publish/MFMA", 25)
timeit("no sweep/publish/MFMA/transc", 27)
timeit("no sweep/publish/transc", 26)
ops.set_option("lstm_ablate", 0)

# phase stamps (diagnostic build path: lstm_ablate bit 5)
ops.set_option("lstm_ablate", 32)
ops.lstm_encode(nets); torch.cuda.synchronize()
ws = ops.workspaces(dev).encode()
prof = ws[32:32 + 56].view(torch.int64).cpu().tolist()
n = max(prof[5], 1)
print("flush + input prefetch part of the MFMA phase:", round(prof[6] / n))
names = ["sweep+LDS fill", "barrier 1", "A-frag reads + MFMA (+flush)", "projection + barrier 2", "cell + publish"]
print("phase cycles per step (wg 0, wave 0):", {k: round(v / n) for k, v in zip(names, prof[:5])}, "steps", n,
      "total", round(sum(prof[:5]) / n))
ops.set_option("lstm_ablate", 0)

ops.set_option("lstm_ablate", 0)
ops.lstm_encode(nets); torch.cuda.synchronize()
ws = ops.workspaces(dev).encode()
print("status", int(ws[:4].view(torch.int32).item()), "workgroups on the same-XCD fast path:", int(ws[4:8].view(torch.int32).item()))
for _ in range(3): ops.lstm_encode(nets, write_through=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.lstm_encode(nets, write_through=True)
e1.record(); torch.cuda.synchronize()
print(f"write-through hand-off forced: {e0.elapsed_time(e1) / 10:.3f} ms")
