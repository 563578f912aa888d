"""Phase stamps of the cooperative decoder (diagnostic option; timing only)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import ops
from gnnpn_sc_amd.modelPN import two_level_greedy
from bench import build_models
dev = torch.device("cuda:0")
T, K, B = 47, 5, 256
net, low, high = build_models(T, 2507, K, dev)
g = torch.Generator().manual_seed(0)
x = torch.rand(B, T * K, 8, generator=g).to(dev)
for _ in range(3): two_level_greedy(low, high, x)
ops.set_option("lstm_ablate", 0x800)
two_level_greedy(low, high, x); torch.cuda.synchronize()
ws = ops.workspaces(dev).decode(B, T, K)
prof = ws[32:32 + 64].view(torch.int64).cpu().tolist()
n = max(prof[7], 1)
names = ["sweep of h: wait", "h to LDS + barrier", "window loads + W_hh.h", "sweep of the partial dots: wait", "logits, argmax, barrier, pick's row",
         "input side + cell + publish h", "barrier + partial dots + publish"]
print("f32  ", {k: round(v / n) for k, v in zip(names, prof[:7])}, "steps", n, "total", round(sum(prof[:7]) / n))
for _ in range(2): two_level_greedy(low, high, x, precision="split")
torch.cuda.synchronize()
ws = ops.workspaces(dev).decode(B, T, K)
prof = ws[32:32 + 64].view(torch.int64).cpu().tolist()
n = max(prof[7], 1)
print("split", {k: round(v / n) for k, v in zip(names, prof[:7])}, "steps", n, "total", round(sum(prof[:7]) / n))
ops.set_option("lstm_ablate", 0)
