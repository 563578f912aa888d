"""Fuzz campaign for the front half (GNN scores -> stable ranking -> candidate reduction) against the oracle on
random shapes, including constraint ranges that leave categories without a feasible service.
Usage: fuzz_front.py [n_configs] [seed]"""
import sys, os, random, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import ml as oml, data as odata, pn as opn      # checker use only
import gnnpn_sc_amd.synth as synth
from gnnpn_sc_amd.modelML import Net
from gnnpn_sc_amd.modelPN import CombinatorialRL, reward
from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline
from gnnpn_sc_amd import ops
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
fails, t0 = 0, time.time()
for c in range(n_cfg):
    T = rng.randint(2, 24); per = rng.randint(1, 30); S = T * per
    K = rng.randint(1, 6); B = rng.choice([1, 2, 7, 16, 19]); n_t = rng.randint(1, min(T, 12))
    n_gcn = rng.choice([1, 2, 4]); deg = rng.randint(1, 12)
    lo = rng.choice([(0.0, 0.5), (0.85, 0.96), (0.93, 0.99), (0.97, 0.999)])
    seeds = [rng.randrange(1 << 20) for _ in range(4)]
    try:
        table = synth.make_service_table(T, S, seed=seeds[0], degree=deg)
        pb = synth.make_problem_batch(table, B, seed=seeds[1], tasks_per_problem=n_t, lo_range=lo)
        sd_ml = oml.make_state_dict(128, 20, 2, n_gcn, seed=seeds[2])
        net = Net(128, S, 20, 2, n_gcn); net.load_state_dict(sd_ml)
        low = CombinatorialRL(0, 32, T * K, 0, 10, 1, reward, "Dot", K, T, level="Low")
        high = CombinatorialRL(0, 32, T * K, 0, 10, 1, reward, "Dot", K, T, level="High")
        low.load_state_dict(opn.make_state_dict(32, 1)); high.load_state_dict(opn.make_state_dict(32, 2))
        pipe = ML2PNPipeline(net.to(dev).eval(), low.to(dev).eval(), high.to(dev).eval(), K)
        svc, batch = DeviceServices.from_table(table, dev), DeviceBatch.from_problems(pb, dev)
        scores = pipe.scores(svc, batch)
        rows, ids = pipe.candidates(svc, batch, scores)
        data = oml.make_data(torch.from_numpy(pb.x), torch.from_numpy(pb.edge_index), torch.from_numpy(pb.batch),
                             torch.from_numpy(table.x_service), torch.from_numpy(table.edge_index),
                             torch.from_numpy(table.edge_attr))
        ref_scores = oml.net_forward(sd_ml, data, 2, n_gcn)
        err = float((scores.cpu() - ref_scores).abs().max())
        assert err < 1e-5, f"scores differ by {err:.2e}"
        ds_like = {str(k + 1): table.qos[table.cat_ptr[k]:table.cat_ptr[k + 1]].tolist() for k in range(T)}
        rank = oml.rank_services(scores.cpu()).tolist()
        assert torch.equal(ops.rank_rows(scores).cpu().long(), torch.tensor(rank)), "ranking"
        nodes_per = pb.x.shape[0] // B
        want, n_dummy = [], 0
        for b in range(B):
            nodes = []
            for r in pb.x[b * nodes_per:(b + 1) * nodes_per]:
                onehot = [0] * (T + 1); onehot[int(r[0])] = 1
                nodes.append(onehot + [float(v) for v in r[1:].astype(np.float64)])
            want.append(odata.reduce_candidates(rank[b], nodes, ds_like, K)[0])
        want = torch.tensor(want, dtype=torch.float32)[:, :, 1:]
        assert torch.equal(rows.cpu(), want), "candidate rows"
        n_dummy = int((ids.cpu() == -1).sum())
        print(f"cfg {c}: T={T} S={S} K={K} B={B} n_t={n_t} gcn={n_gcn} lo={lo} score err {err:.1e} dummy rows {n_dummy} ok", flush=True)
    except Exception as e:          # noqa: BLE001
        fails += 1
        print(f"FAIL cfg {c}: T={T} S={S} K={K} B={B} n_t={n_t} gcn={n_gcn} deg={deg} lo={lo} seeds={seeds}: {type(e).__name__} {e}")
print(f"{n_cfg} configs, {fails} failures, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
