import sys, os, torch
sys.path.insert(0, os.getcwd())
from gnnpn_sc_amd import ops
dev = torch.device("cuda:0")
B, L, H = 512, 235, 256
g = torch.Generator().manual_seed(0)
x = torch.rand(B, L, 8, generator=g).to(dev)
nets = [{"inputs": x, "w_in": ((torch.rand(4 * H, 8, generator=g) * 2 - 1) * 0.3).to(dev), "b_in": ((torch.rand(4 * H, generator=g) * 2 - 1) * 0.3).to(dev),
         "whh": ops.pack_lstm_weight((torch.rand(4 * H, H, generator=g) * 2 - 1) / 16).to(dev), "bhh": ((torch.rand(4 * H, generator=g) * 2 - 1) / 16).to(dev)} for _ in range(2)]
for _ in range(3): ops.lstm_encode(nets, precision="split", impl=5)
ops.set_option("lstm_ablate", 32)
ops.lstm_encode(nets, precision="split", impl=5); torch.cuda.synchronize()
ws = ops.workspaces(dev).encode()
prof = ws[32:32 + 48].view(torch.int64).cpu().tolist()
ops.set_option("lstm_ablate", 0)
n = max(prof[4], 1)
print("pair kernel, cycles per tile-step:", {k: round(prof[i] / n) for i, k in enumerate(["input MFMAs + sweep", "barrier", "chain", "prefetch+flush+cell+publish"])}, "tile-steps", n, "sum", round(sum(prof[:4]) / n))
