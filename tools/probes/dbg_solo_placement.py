"""Solo encoder launches (one stream, nothing beside them): per-launch time and the launch's own placement counters."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from gnnpn_sc_amd import ops
dev = torch.device("cuda:0")
H, B, L = 256, 256, 235
g = torch.Generator().manual_seed(0)
x = torch.rand(B, L, 8, generator=g).to(dev)
nets = [{"inputs": x, "w_in": ((torch.rand(4 * H, 8, generator=g) * 2 - 1) * 0.3).to(dev), "b_in": ((torch.rand(4 * H, generator=g) * 2 - 1) * 0.3).to(dev),
         "whh": ops.pack_lstm_weight((torch.rand(4 * H, H, generator=g) * 2 - 1) / 16).to(dev), "bhh": ((torch.rand(4 * H, generator=g) * 2 - 1) / 16).to(dev)} for _ in range(2)]
for lds_kb in (0, 78):
    for _ in range(5):
        ops.lstm_encode(nets, precision="split", lds_kb=lds_kb)
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.lstm_encode(nets, precision="split", lds_kb=lds_kb)
        e1.record()
        torch.cuda.synchronize()
        ts.append(round(e0.elapsed_time(e1) * 100, 1))
    ws = ops.workspaces(dev)
    v = ws.encode().view(torch.int32).cpu()
    claims = v[2048 // 4: 2048 // 4 + 8 * 256].view(8, 256)
    print(f"lds_kb {lds_kb}: us per launch {ts}; last launch: {ws.placement()}, words0_7 {[int(t) for t in v[:8]]}, "
          f"workgroups that reached a CU: per XCD max {[int(claims[i].max()) for i in range(8)]} sum {int(claims.sum())}", flush=True)
