import sys, os, torch
sys.path.insert(0, os.getcwd())
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
# prof[7..9] of workgroup 0: placement (entry -> seated), weights into registers, first step — shader cycles
for prec in ("split+presplit", "split", "f32"):
    if prec.endswith("+presplit"):
        for d in nets: d["whh_split"] = ops.pack_lstm_split_weights(d["whh"])
    else:
        for d in nets: d.pop("whh_split", None)
    prec = prec.split("+")[0]
    for _ in range(3): ops.lstm_encode(nets, precision=prec)
    ops.set_option("lstm_ablate", 32)
    for rep in range(2):
        ops.lstm_encode(nets, precision=prec); torch.cuda.synchronize()
        ws = ops.workspaces(dev).encode()
        prof = ws[32:32 + 80].view(torch.int64).cpu().tolist()
        print(prec, rep, prof[:10], flush=True)
    ops.set_option("lstm_ablate", 0)
