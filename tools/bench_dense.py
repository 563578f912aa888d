"""The dense layers of the GIN branch at the 1000-task shapes, one by one (M = 512 x 1001 node rows): time, TFLOP/s, GB/s.
    python tools/bench_dense.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import ops
dev = torch.device("cuda:0")
M = 512 * 1001
g = torch.Generator().manual_seed(0)
for K, N, epi in ((26, 256, True), (256, 128, True), (128, 256, True), (256, 128, True), (128, 128, False)):
    a = torch.rand(M, K, generator=g).to(dev)
    w = ((torch.rand(N, K, generator=g) - 0.5) * 0.2).to(dev)
    b = torch.rand(N, generator=g).to(dev)
    sc, sh = (torch.rand(N, generator=g).to(dev), torch.rand(N, generator=g).to(dev)) if epi else (None, None)
    out = torch.empty(M, N, device=dev)
    f = lambda: ops.linear(a, w, b, sc, sh, ops.ACT_RELU if epi else ops.ACT_NONE, out=out)   # noqa: E731
    for _ in range(3):
        f()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    flop, byt = 2.0 * M * N * K, 4.0 * M * (K + N)
    print(f"[{M} x {K}] x [{N} x {K}]^T{' +BN+ReLU' if epi else ''}: {best * 1e3:7.1f} us  {flop / best / 1e9:6.1f} TFLOP/s (fp32 matrix peak 157)  "
          f"{byt / best / 1e6:7.1f} GB/s  floors: MFMA {flop / 157.3e12 * 1e6:5.1f} us, HBM {byt / 8e12 * 1e6:5.1f} us", flush=True)
