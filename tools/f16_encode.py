"""fp16-operand encoder (opt-in) vs fp32: timing and agreement on a QWS-shaped two-level decode."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import ops
from gnnpn_sc_amd.modelPN import two_level_greedy
from bench import build_models
dev = torch.device("cuda:0")
T, K, B = 47, 5, 256
net, low, high = build_models(T, 2507, K, dev)
g = torch.Generator().manual_seed(0)
x = torch.rand(B, T * K, 8, generator=g)
x[:, :, 2:4] = 0.9 + 0.1 * x[:, :, 2:4]
x[:, K:, 4:] = 0
x = x.to(dev)
def run(prec):
    prec = {0: "f32", 1: "f16", 2: "split"}[prec]
    for _ in range(3): out = two_level_greedy(low, high, x, precision=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = two_level_greedy(low, high, x, precision=prec)
    e1.record(); torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / 10
a, ta = run(0)
b, tb = run(1)
a, ta2 = run(0)
b, tb2 = run(1)
print("repeat:", round(ta2, 3), round(tb2, 3))
ops.check_status(dev)
same_low = (a["idx_low"] == b["idx_low"]).all(1).float().mean().item()
same_high = (a["idx_high"] == b["idx_high"]).all(1).float().mean().item()
dec = (a["idx_high"] == b["idx_high"]).float().mean().item()
print(f"fp32 {ta:.3f} ms, fp16-encoder {tb:.3f} ms per two-level pass (B={B})")
print(f"problems with identical Low picks {same_low:.3f}, identical High picks {same_high:.3f}; decisions identical {dec:.4f}")
print("max |win_low diff|", float((a["win_low"] - b["win_low"]).abs().max()), " mean |R diff|", float((a["R"] - b["R"]).abs().mean()),
      " mean R fp32", float(a["R"].mean()), " mean R fp16", float(b["R"].mean()))
