"""Encoder precision modes (f32 / split / f16): error of enc_out against an fp64 CPU LSTM, timing, and
agreement of the two-level decode with the f32 path.  Diagnostic; not part of the product path."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import ops
from gnnpn_sc_amd.modelPN import two_level_greedy
from bench import build_models
dev = torch.device("cuda:0")
T, K, B = int(os.environ.get("SE_T", 47)), int(os.environ.get("SE_K", 5)), int(os.environ.get("SE_B", 256))   # SE_T=1000 SE_B=32: L = 5000
net, low, high = build_models(T, 2507 if T == 47 else 5 * T, K, dev)
scale = float(os.environ.get("WHH_SCALE", "1"))          # stress: larger recurrent weights (saturating gates)
if scale != 1:
    with torch.no_grad():
        for m in (low, high):
            m.actor.encoder.weight_hh_l0.mul_(scale); m.actor.decoder.weight_hh_l0.mul_(scale)
            m.actor.encoder.weight_ih_l0.mul_(scale)
    print("W_hh, W_ih scaled by", scale)
g = torch.Generator().manual_seed(0)
x = torch.rand(B, T * K, 8, generator=g)
x[:, :, 2:4] = 0.9 + 0.1 * x[:, :, 2:4]
x[:, K:, 4:] = 0
xd = x.to(dev)

# fp64 reference of the Low encoder on the first 32 problems
sd = {k: v.detach().cpu().double() for k, v in low.state_dict().items()}
nb = min(32, B)
emb = x[:nb].double() @ sd["actor.embedding2.weight"].T + sd["actor.embedding2.bias"]
lstm = torch.nn.LSTM(256, 256, batch_first=True).double()
lstm.load_state_dict({k.replace("actor.encoder.", ""): v for k, v in sd.items() if k.startswith("actor.encoder.")})
with torch.no_grad():
    ref, _ = lstm(emb)

res = {}
for name in ("f32", "split", "f16"):
    for _ in range(2): out = two_level_greedy(low, high, xd, precision=name)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = two_level_greedy(low, high, xd, precision=name)
    e1.record(); torch.cuda.synchronize()
    res[name] = (out, e0.elapsed_time(e1) / 10)
    # encoder alone
    from gnnpn_sc_amd import modelPN
    el, eh = low.actor.encode_args(xd)[0], high.actor.encode_args(xd)[0]
    for _ in range(2): enc, h_n, c_n = ops.lstm_encode([el, eh], precision=name)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10): enc, h_n, c_n = ops.lstm_encode([el, eh], precision=name)
    e1.record(); torch.cuda.synchronize()
    err = (enc[0][:nb].double().cpu() - ref).abs()
    print(f"L={T * K} {name:6s} encoder {e0.elapsed_time(e1) / 10:.3f} ms   two-level pass {res[name][1]:.3f} ms   "
          f"enc_out vs fp64: max {err.max():.3e} mean {err.mean():.3e}  (last step max {err[:, -1].max():.3e})")
ops.check_status(dev)
a = res["f32"][0]
for name in ("split", "f16"):
    b = res[name][0]
    same = (a["actions"] == b["actions"]).all(-1)
    print(f"{name:6s} vs f32: problems with identical selection {same.all(1).float().mean():.4f}, decisions {same.float().mean():.5f}, "
          f"max |win_low diff| {float((a['win_low'] - b['win_low']).abs().max()):.3e}, mean |R diff| {float((a['R'] - b['R']).abs().mean()):.2e}")
