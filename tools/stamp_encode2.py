import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import ops
dev = torch.device("cuda:0")
B, L, H = 256, 235, 256
g = torch.Generator().manual_seed(0)
x = torch.rand(B, L, 8, generator=g).to(dev)
nets = [{"inputs": x, "w_in": ((torch.rand(4 * H, 8, generator=g) * 2 - 1) * 0.3).to(dev),
         "b_in": ((torch.rand(4 * H, generator=g) * 2 - 1) * 0.3).to(dev),
         "whh": ops.pack_lstm_weight((torch.rand(4 * H, H, generator=g) * 2 - 1) / 16).to(dev),
         "bhh": ((torch.rand(4 * H, generator=g) * 2 - 1) / 16).to(dev)} for n in range(2)]
ops.set_option("lstm_impl", 3)
for _ in range(3): ops.lstm_encode(nets)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.lstm_encode(nets); e1.record(); torch.cuda.synchronize(); print("coop2 ms", e0.elapsed_time(e1))
ops.set_option("lstm_ablate", 32)
ops.lstm_encode(nets); torch.cuda.synchronize()
ws = ops.encode_workspace(dev)
names = ["sweep+fill", "4-wave barrier", "flush+A-frag+64 MFMA", "proj+cell+publish"]
for h in range(2):
    prof = ws[32 + h * 64: 32 + h * 64 + 40].view(torch.int64).cpu().tolist()
    n = max(prof[4], 1)
    print("half", h, {k: round(v / n) for k, v in zip(names, prof[:4])}, "steps", n, "total", round(sum(prof[:4]) / n))
ops.set_option("lstm_ablate", 0)
ops.set_option("lstm_impl", 0)
