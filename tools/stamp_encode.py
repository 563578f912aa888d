"""Phase stamps of the cooperative encoder (diagnostic build; timing only), fp32 and exact split: cycles per step."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import ops
dev = torch.device("cuda:0")
B, L, H = int(os.environ.get("SE_B", 256)), int(os.environ.get("SE_L", 235)), 256
g = torch.Generator().manual_seed(0)
x = torch.rand(B, L, 8, generator=g).to(dev)
nets = []
for n in range(2):
    nets.append({"inputs": x, "w_in": ((torch.rand(4 * H, 8, generator=g) * 2 - 1) * 0.3).to(dev),
                 "b_in": ((torch.rand(4 * H, generator=g) * 2 - 1) * 0.3).to(dev),
                 "whh": ops.pack_lstm_weight((torch.rand(4 * H, H, generator=g) * 2 - 1) / 16).to(dev),
                 "bhh": ((torch.rand(4 * H, generator=g) * 2 - 1) / 16).to(dev)})
names = ["input copy + hand-off sweep + LDS fill", "barrier", "prefetch issue + enc_out flush + A-fragments + MFMAs", "input projection + barrier",
         "cell update + publish", None, "(of the MFMA phase: prefetch issue + flush)"]
for prec in ("f32", "split"):
    for _ in range(3): ops.lstm_encode(nets, precision=prec)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops.lstm_encode(nets, precision=prec)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(prec, f"B={B} L={L}: {ms:.3f} ms per launch (production build), {ms / L * 1e3:.3f} us per step", flush=True)
    ops.set_option("lstm_ablate", 32)
    ops.lstm_encode(nets, precision=prec); torch.cuda.synchronize()
    ws = ops.workspaces(dev).encode()
    prof = ws[32:32 + 80].view(torch.int64).cpu().tolist()
    ops.set_option("lstm_ablate", 0)
    n = max(prof[5], 1)
    print(prec, {k: round(prof[i] / n) for i, k in enumerate(names) if k}, "steps", n, "total", round(sum(prof[:5]) / n), flush=True)
    print(prec, "fixed part of the launch (workgroup 0, cycles): placement", prof[7], "weights into registers", prof[8],
          "first step (no stamps of its own)", prof[9], flush=True)
