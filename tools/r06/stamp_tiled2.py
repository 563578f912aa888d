"""Phase stamps of the two-buffer tiled aggregate (round 6, diagnostic): the kernel writes s_memtime at every phase boundary of its
third and fourth item, per wavefront, into the buffer named by GNNPN_TILED2_STAMPS.  python tools/r06/stamp_tiled2.py [S:copies] [dbg]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gnnpn_sc_amd import graph, ops, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "5000:128"
if len(sys.argv) > 2:
    os.environ["GNNPN_TILED2_DBG"] = sys.argv[2]
S, copies = (int(v) for v in cfg.split(":"))
dev = torch.device("cuda:0")
table = synth.make_service_table(47, S, 0, degree=32, graph="scan")
csr = graph.gcn_csr(torch.from_numpy(table.edge_index), torch.from_numpy(table.edge_attr), S)
nnz = csr.col.numel()
rp = torch.cat([csr.rowptr[:-1].long() + c * nnz for c in range(copies)] + [torch.tensor([copies * nnz])]).int().to(dev)
col = torch.cat([csr.col.long() + c * S for c in range(copies)]).int().to(dev)
norm = ops.gcn_norm(rp, col, csr.w.repeat(copies).to(dev))
x = torch.randn(copies * S, 256, device=dev)
bias = torch.randn(256, device=dev)
scale, shift = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev)
plan = ops.TilePlan(rp, col, norm, S, form=1)
NT = plan.geom["src_tiles"]
fn = lambda: plan.aggregate(x, None, bias, scale, shift, ops.ACT_RELU)   # noqa: E731
for _ in range(20):
    fn()
torch.cuda.synchronize()
st = torch.zeros(256 * 16 * 64, dtype=torch.int64, device=dev)
os.environ["GNNPN_TILED2_STAMPS"] = hex(st.data_ptr())
fn()
torch.cuda.synchronize()
del os.environ["GNNPN_TILED2_STAMPS"]
a = st.cpu().numpy().reshape(256, 16, 2, 32)
out = {"config": cfg, "dbg": os.environ.get("GNNPN_TILED2_DBG", "0"), "src_tiles": NT}
g = a[:, :15, :, :].astype(np.float64)                      # gatherers
ok = g[:, :, :, 0] > 0
def med(v):
    v = v[ok]
    return [round(float(np.percentile(v, q))) for q in (10, 50, 90)]
out["item_cycles_p10_p50_p90"] = med(g[..., 28] - g[..., 0])
out["start_to_first_barrier"] = med(g[..., 1] - g[..., 0])
for t in range(NT):
    out[f"tile{t}_wait_at_barrier"] = med(g[..., 2 + 3 * t] - g[..., 1 + 3 * t])
    out[f"tile{t}_gather"] = med(g[..., 3 + 3 * t] - g[..., 2 + 3 * t])
out["epilogue"] = med(g[..., 28] - g[..., 3 * NT])
# per workgroup: spread of the gather end over its wavefronts (what the next barrier waits for)
spread = []
for t in range(NT):
    e = g[..., 3 + 3 * t]
    spread.append(round(float(np.median((e.max(axis=1) - e.min(axis=1))[ok.all(axis=1)]))))
out["gather_end_spread_over_wavefronts_median"] = spread
l = a[:, 15, :, :].astype(np.float64)
okl = l[:, :, 0] > 0
for t in range(NT):
    out[f"loader_tile{t}"] = {k: [round(float(np.percentile(v[okl], q))) for q in (10, 50, 90)] for k, v in
                              (("issue", l[..., 4 * t + 1] - l[..., 4 * t]), ("landed_after_issue", l[..., 4 * t + 2] - l[..., 4 * t + 1]),
                               ("barrier", l[..., 4 * t + 3] - l[..., 4 * t + 2]))}
print(json.dumps(out))
