"""Solo timing of the decoder forms (eager, one stream)."""
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
import gnnpn_sc_amd.ops as O
orig = O.pointer_decode
for impl in (0, 3, 2, 4, 3, 4):
    ops.set_option("decode_impl", impl)
    for _ in range(3): two_level_greedy(low, high, x)
    torch.cuda.synchronize()
    evs = []
    def timed(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(*a, **k); e1.record(); evs.append((e0, e1)); return r
    O.pointer_decode = timed
    for _ in range(10): two_level_greedy(low, high, x)
    torch.cuda.synchronize()
    O.pointer_decode = orig
    print("decode_impl", impl, "avg ms", sum(a.elapsed_time(b) for a, b in evs) / len(evs))
ops.set_option("decode_impl", 0)
ops.check_status(dev)
