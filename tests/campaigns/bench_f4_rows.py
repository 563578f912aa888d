"""Timings of the SURVEY 8f row 4 paths next to the CPU oracle (torch restatements of the reference):
  * one training step of the GNN model (trainML.py:39-45) at the QWS shape, batches of two graphs (trainML.py:121);
  * the two-level greedy decode with 'Bahdanau' attention and one glimpse round at the QWS shape, 128 problems.
python tests/campaigns/bench_f4_rows.py   -> one JSON line per row (committed as profiles/r02_f4_rows.jsonl)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from pn_inputs import pn_inputs
from oracle import ml as oml, ml_train as omt, pn as opn
from gnnpn_sc_amd import synth
from gnnpn_sc_amd.modelML import Net
from gnnpn_sc_amd.modelPN import CombinatorialRL, reward, two_level_greedy
from gnnpn_sc_amd.trainML import MLAdam, MLBatch, ml_train_step
dev = torch.device("cuda:0")

# ---- GNN training step
T, S, hidden, emb, n_gin, n_gcn = 47, 2507, 128, 20, 2, 2
table = synth.make_service_table(T, S, seed=0, degree=32)
pb = synth.make_problem_batch(table, 2, seed=1, tasks_per_problem=10)
sd = oml.make_state_dict(hidden, emb, n_gin, n_gcn, 3)
net = Net(hidden, S, emb, n_gin, n_gcn)
net.load_state_dict(sd)
net = net.to(dev).train()
y = (torch.rand(2 * S, generator=torch.Generator().manual_seed(4)) < 0.004).float()
n0 = int((pb.batch == 0).sum())
m0 = pb.batch[pb.edge_index[0]] == 0
graphs = [{"x": torch.from_numpy(pb.x[:n0]), "edge_index": torch.from_numpy(pb.edge_index[:, m0]), "y": y[:S]},
          {"x": torch.from_numpy(pb.x[n0:]), "edge_index": torch.from_numpy(pb.edge_index[:, ~m0] - n0), "y": y[S:]}]
service = {"x_service": torch.from_numpy(table.x_service), "edge_index_service": torch.from_numpy(table.edge_index),
           "edge_attr_service": torch.from_numpy(table.edge_attr)}
mb = MLBatch(graphs, service, dev)                       # first call: CUDA context, allocator warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    mb = MLBatch(graphs, service, dev)
torch.cuda.synchronize()
batch_ms = (time.perf_counter() - t0) / 5 * 1e3
adam = MLAdam(net, 1e-3)
for _ in range(3):
    ml_train_step(net, mb, adam)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    loss = ml_train_step(net, mb, adam)
torch.cuda.synchronize()
step_ms = (time.perf_counter() - t0) / n * 1e3
torch.set_num_threads(16)
nodes = [n0]
ei, ea = oml.pyg_batch_service_edges(service["edge_index_service"], service["edge_attr_service"], [0, n0])
data = oml.make_data(torch.from_numpy(pb.x), torch.from_numpy(pb.edge_index), torch.from_numpy(pb.batch),
                     service["x_service"].repeat(2, 1), ei, ea)
omt.train_step(sd, data, y, n_gin, n_gcn, S, 1e-3)
t0 = time.perf_counter()
omt.train_step(sd, data, y, n_gin, n_gcn, S, 1e-3)
cpu_ms = (time.perf_counter() - t0) * 1e3
print(json.dumps({"row": "GNN training step (trainML.py:39-45)", "shape": f"QWS: S={S}, 2 graphs per batch, {table.edge_index.shape[1]} service edges",
                  "hip_step_ms": round(step_ms, 2), "batch_layout_ms_host": round(batch_ms, 1), "loss": round(float(loss.item()), 5),
                  "cpu_oracle_step_ms_16_threads": round(cpu_ms, 1)}), flush=True)

# ---- attention forms
T, K, B, H = 47, 5, 128, 256
x = pn_inputs(B, T, K, 3)
for att, ng in (("Dot", 1), ("Bahdanau", 1)):
    nets = []
    sds = []
    for lvl, seed in (("Low", 1), ("High", 2)):
        m = CombinatorialRL(0, H, T * K, ng, 10, 1, reward, att, K, T, level=lvl)
        sds.append(opn.make_state_dict(H, seed, attention=att))
        m.load_state_dict(sds[-1])
        nets.append(m.to(dev).eval())
    xd = x.to(dev)
    for _ in range(2):
        out = two_level_greedy(nets[0], nets[1], xd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = two_level_greedy(nets[0], nets[1], xd)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    t0 = time.perf_counter()
    ref = opn.two_level_greedy(sds[0], sds[1], x, T, K, attention=att, n_glimpses=ng)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    same = float((out["idx_high"].cpu().long() == ref["idx_high"]).all(1).float().mean())
    print(json.dumps({"row": f"two-level greedy decode, attention={att}, n_glimpses={ng}", "shape": f"QWS: B={B} T={T} K={K} H={H}",
                      "hip_ms": round(ms, 2), "problems_per_s": round(B / ms * 1e3, 1), "cpu_oracle_ms_16_threads": round(cpu_ms, 1),
                      "problems_identical_to_oracle": same}), flush=True)
