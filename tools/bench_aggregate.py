"""HBM roofline of the CSR gather-aggregate kernel on the reference's own batching of the service
graph: a PyG batch of B problems holds B block-diagonal copies of the service graph
(trainML.py:109-114, modelML.py:145-156), i.e. one GCN aggregate over B*S rows.  The product path
evaluates the service branch once per forward; this "replicated" mode exists only to
load the kernel to the point where HBM, not launch latency, is the bound.

    python tools/bench_aggregate.py [--S 2507] [--copies 256] [--degree 32]

Algorithmic bytes per launch: 2*N*C*4 + E*(4+4) + (N+1)*4, N = copies*S, C = 256.
"""
import argparse, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import graph, ops, synth

ap = argparse.ArgumentParser()
ap.add_argument("--S", type=int, default=2507)
ap.add_argument("--copies", type=int, default=256)
ap.add_argument("--degree", type=int, default=32)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
table = synth.make_service_table(47, a.S, 0, degree=a.degree)
csr = graph.gcn_csr(torch.from_numpy(table.edge_index), torch.from_numpy(table.edge_attr), a.S)
nnz = csr.col.numel()
# block-diagonal replication: rowptr/col offset per copy
rp = torch.cat([csr.rowptr[:-1].long() + c * nnz for c in range(a.copies)] + [torch.tensor([a.copies * nnz])]).int()
col = torch.cat([csr.col.long() + c * a.S for c in range(a.copies)]).int()
w = csr.w.repeat(a.copies)
N, C = a.copies * a.S, 256
rp, col, w = rp.to(dev), col.to(dev), w.to(dev)
norm = ops.gcn_norm(rp, col, w)
x = torch.randn(N, C, device=dev)
bias = torch.randn(C, device=dev)
scale, shift = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
for _ in range(3):
    y = ops.csr_aggregate(rp, col, norm, x, bias=bias, scale=scale, shift=shift, act=ops.ACT_RELU)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    y = ops.csr_aggregate(rp, col, norm, x, bias=bias, scale=scale, shift=shift, act=ops.ACT_RELU)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
alg = 2 * N * C * 4 + col.numel() * 8 + (N + 1) * 4
# one copy checked against the single-graph result (bit-exact: same CSR order)
y1 = ops.csr_aggregate(csr.rowptr.to(dev), csr.col.to(dev), norm[:nnz].contiguous(), x[:a.S].contiguous(), bias=bias,
                       scale=scale, shift=shift, act=ops.ACT_RELU)
assert torch.equal(y[:a.S], y1)
print(json.dumps({"kernel": "csr_aggregate_kernel<true> (GCN layer, bias+BN+ReLU epilogue)", "rows": N, "channels": C,
                  "nnz": int(col.numel()), "ms": round(ms, 4), "algorithmic_bytes": alg,
                  "achieved_GBps": round(alg / ms / 1e6, 1), "peak_GBps": 8000.0,
                  "frac_of_spec": round(alg / ms / 1e6 / 8000.0, 4), "frac_of_measured_copy_6290": round(alg / ms / 1e6 / 6290.0, 4)}))
