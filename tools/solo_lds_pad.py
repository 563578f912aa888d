"""Solo cooperative kernels with and without the LDS padding that forces one workgroup per CU."""
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
el, eh = low.actor.encode_args(x)[0], high.actor.encode_args(x)[0]
def t_enc(prec):
    for _ in range(3): ops.lstm_encode([el, eh], precision=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.lstm_encode([el, eh], precision=prec)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
def t_pass(prec):
    for _ in range(3): two_level_greedy(low, high, x, precision=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): two_level_greedy(low, high, x, precision=prec)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
for kb in (0, 100, 0, 100):
    ops.set_option("coop_lds_kb", kb)
    for impl in (0, 3):
        ops.set_option("decode_impl", impl)
        print(f"coop_lds_kb={kb:3d} decode_impl={impl}: encoder f32 {t_enc('f32'):.4f} ms, split {t_enc('split'):.4f} ms; "
              f"two-level pass f32 {t_pass('f32'):.4f} ms, split {t_pass('split'):.4f} ms")
ops.set_option("coop_lds_kb", 0); ops.set_option("decode_impl", 0)
ops.check_status(dev)
