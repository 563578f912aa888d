import os, sys, ctypes, json
os.environ["GNNPN_LIB"] = "gnnpn-sc_amd/build/ablate/libgnnpn_hip_aggregate_stamps_vw.so"
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gnnpn_sc_amd import _lib, graph, ops, synth
dev = torch.device("cuda:0")
for S, copies in ((5000, 128), (2507, 256)):
    table = synth.make_service_table(47, S, 0, degree=32, graph="scan")
    csr = graph.gcn_csr(torch.from_numpy(table.edge_index), torch.from_numpy(table.edge_attr), S)
    nnz = csr.col.numel()
    rp = torch.cat([csr.rowptr[:-1].long() + c * nnz for c in range(copies)] + [torch.tensor([copies * nnz])]).int().to(dev)
    col = torch.cat([csr.col.long() + c * S for c in range(copies)]).int().to(dev)
    norm = ops.gcn_norm(rp, col, csr.w.repeat(copies).to(dev))
    x = torch.randn(copies * S, 256, device=dev)
    plan = ops.csr_tile_plan(rp, col, norm, S)
    NT = plan.geom["src_tiles"]
    for _ in range(10): plan.aggregate(x)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 16 * 32, np.uint64)
    lib = _lib.load()
    lib.gnnpn_debug_agg_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    lib.gnnpn_debug_agg_stamps(buf.ctypes.data, buf.nbytes)
    st = buf.reshape(4096, 16, 32).astype(np.int64)
    n = 5 + 3 * NT
    gather = sum(st[:, :, 4 + 3 * t] - st[:, :, 3 + 3 * t] for t in range(NT))     # per wave: own units done - barrier passed
    print(json.dumps({"S": S, "copies": copies, "per_wave_mean": {"gather_cycles": float(gather.mean()), "waiting_for_the_edge_list_loads": float(st[:, :, 29].mean()),
          "iterations": float(st[:, :, 28].mean()), "wait_per_iteration": float(st[:, :, 29].sum() / st[:, :, 28].sum())}}))
