"""Fixed cost of one cooperative encoder / decoder launch: time per launch against the sequence length, B = 256 (one tile per
group); the intercept of the line is what a launch costs before its first step and after its last (launch, placement, weights
into registers, the epilogue), the slope the step.   python tools/probes/encode_intercept.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnnpn_sc_amd import ops
dev = torch.device("cuda:0")
B, H = 256, 256
g = torch.Generator().manual_seed(0)
w = [{"w_in": ((torch.rand(4 * H, 8, generator=g) * 2 - 1) * 0.3).to(dev), "b_in": ((torch.rand(4 * H, generator=g) * 2 - 1) * 0.3).to(dev),
      "whh": ops.pack_lstm_weight((torch.rand(4 * H, H, generator=g) * 2 - 1) / 16).to(dev),
      "bhh": ((torch.rand(4 * H, generator=g) * 2 - 1) / 16).to(dev)} for _ in range(2)]
for prec in ("split", "f32"):
    pts = []
    for L in (5, 20, 50, 100, 235, 470, 940):
        x = torch.rand(B, L, 8, generator=g).to(dev)
        nets = [dict(n, inputs=x) for n in w]
        for _ in range(3):
            ops.lstm_encode(nets, precision=prec)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.lstm_encode(nets, precision=prec)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 4 * 1e3)
        pts.append((L, best))
    ops.check_status(dev)
    (l0, t0), (l1, t1) = pts[-3], pts[-1]
    slope = (t1 - t0) / (l1 - l0)
    print(prec, " ".join(f"L={l}: {t:.1f} us" for l, t in pts), f"| slope {slope:.3f} us/step, intercept {t0 - slope * l0:.1f} us (launch incl. its zeroing kernel)", flush=True)
