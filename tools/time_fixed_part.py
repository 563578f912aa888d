import sys, os, torch, json
sys.path.insert(0, os.getcwd())
from gnnpn_sc_amd import ops
dev = torch.device("cuda:0")
H = 256
g = torch.Generator().manual_seed(0)
def nets_for(B, L):
    x = torch.rand(B, L, 8, generator=g).to(dev)
    out = []
    for n in range(2):
        out.append({"inputs": x, "w_in": ((torch.rand(4 * H, 8, generator=g) * 2 - 1) * 0.3).to(dev),
                    "b_in": ((torch.rand(4 * H, generator=g) * 2 - 1) * 0.3).to(dev),
                    "whh": ops.pack_lstm_weight((torch.rand(4 * H, H, generator=g) * 2 - 1) / 16).to(dev),
                    "bhh": ((torch.rand(4 * H, generator=g) * 2 - 1) / 16).to(dev)})
    return out
def timed(fn, reps=30):
    for _ in range(5): fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
for prec in ("split+presplit", "split", "f32"):      # "+presplit": the weights split once per model (ops.pack_lstm_split_weights, round 6)
    rec = {}
    for L in (1, 2, 5, 20, 60, 235):
        nets = nets_for(256, L)
        if prec.endswith("+presplit"):
            for d in nets:
                d["whh_split"] = ops.pack_lstm_split_weights(d["whh"])
        rec[L] = round(timed(lambda: ops.lstm_encode(nets, precision=prec.split("+")[0])) * 1e3, 1)
    # least squares: t = a + b L
    Ls = [20, 60, 235]; ts = [rec[l] for l in Ls]
    b = (ts[2] - ts[0]) / (Ls[2] - Ls[0]); a = ts[2] - b * Ls[2]
    print(json.dumps({"encoder": prec, "us_by_L": rec, "us_per_step": round(b, 3), "fixed_us (incl. zeroing kernel + launch)": round(a, 1)}), flush=True)
