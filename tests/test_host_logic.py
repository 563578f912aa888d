"""-m "not gpu": host-side mirrors (loadData, loadDataPN, ML2PN.check, main.py CLI), layout helpers,
state_dict compatibility, sharding + the gloo world_size-2 all-gather."""
import contextlib
import io
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from oracle import ml as oml
from oracle import pn as opn


@pytest.fixture()
def fx():
    with open(os.path.join(GOLDEN, "data_small.json")) as f:
        return json.load(f)


@pytest.fixture()
def workdir(fx, tmp_path, monkeypatch):
    import gnnpn_sc_amd.synth as synth
    synth.write_dataset(str(tmp_path), "QWS", fx["dataset"])
    (tmp_path / "solutions" / "pretrained").mkdir(parents=True)
    monkeypatch.chdir(tmp_path)
    return tmp_path


def test_loadData_matches_reference(fx, workdir):
    from gnnpn_sc_amd.loadData import loadData
    nodes, services, ei, eis, eas, labels, inv = loadData("QWS")
    g = fx["loadData"]
    assert nodes == g["nodefeatures"] and services == g["serviceFeatureList"]
    assert eis == g["edge_indices_service"] and eas == g["edge_attrs_service"]
    assert ei == fx["dataset"]["edge_indices"] and labels == fx["dataset"]["labels"]
    assert inv.shape == (fx["S"],)


def test_loadDataPN_matches_reference(fx, workdir):
    from gnnpn_sc_amd.loadData import loadDataPN
    P, K, T = fx["P"], fx["K"], fx["T"]
    (workdir / "solutions" / "pretrained" / "QWS-ML.txt").write_text(json.dumps([fx["rank_shared"]] * P))
    rows, labels = loadDataPN(epoch=-1, dataset="QWS", serviceNumber=K)
    assert rows == fx["rows_shared"] and labels == fx["dataset"]["minCostList"]
    # epoch >= 0 reads the other artefact path (loadData.py:84-86)
    (workdir / "solutions" / "ML" / "QWS").mkdir(parents=True)
    (workdir / "solutions" / "ML" / "QWS" / "testServices-epoch2.txt").write_text(json.dumps(fx["rank_each"]))
    rows_e, _ = loadDataPN(epoch=2, dataset="QWS", serviceNumber=K)
    for p in range(P):
        for c in range(T):
            got = sorted({tuple(r) for r in rows_e[p][c * K:(c + 1) * K]})
            assert got == [tuple(r) for r in fx["rows_each_sets"][p][c]]


def test_check_and_cli_match_reference(fx, workdir):
    from gnnpn_sc_amd import ML2PN
    P, T = fx["P"], fx["T"]
    (workdir / "solutions" / "pretrained" / "QWS-ML.txt").write_text(json.dumps([fx["rank_shared"]] * P))
    (workdir / "solutions" / "pretrained" / "QWS-PNHigh.txt").write_text(json.dumps(fx["check"]["actions"]))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = ML2PN.check("QWS", T, -1)
    assert abs(res - fx["check"]["score"]) < 1e-12
    assert buf.getvalue() == fx["check"]["printed"]            # same printed line as the reference
    # CLI: section [QWS-ML+2PN] says serviceCategory = 47; this dataset has T = 6 -> use a local ini
    (workdir / "environment.ini").write_text("[QWS-ML+2PN]\nserviceCategory = 6\nepoch = -1\n"
                                             "[Normal-ML+2PN]\nserviceCategory = 6\nepoch = -1\n")
    import main as cli
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert cli.main(["main.py", "qws", "ML+2PN"]) == 0
    assert buf.getvalue() == fx["check"]["printed"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert cli.main(["main.py", "QWS", "DAAGA"]) == 1
    assert buf.getvalue().strip() == "Please check the parameters!"       # reference main.py:231


def test_infer_weights_follow_the_epoch(tmp_path, monkeypatch):
    """main._models: `main.py <ds> ML+2PN <epoch> --infer` must run on the weights OF THAT EPOCH — the files
    trainML / trainPNHigh write — and must refuse to run on random weights when one is missing (unless --random-init)."""
    import configparser
    import torch
    import main as cli
    from oracle import ml as oml, pn as opn
    monkeypatch.chdir(tmp_path)
    T, S, K, H = 6, 60, 3, 32
    cfg = configparser.RawConfigParser()
    cfg.read_string("[QWS-ML]\nnumLayersGIN = 2\nnumLayersGCN = 2\nhiddenChannels = 128\nembeddingChannels = 20\n"
                    f"[QWS-PNHigh]\nserNumber = {K}\nhidden_size = {H}\nn_glimpses = 0\ntanh_exploration = 10\nuse_tanh = 1\n")
    with pytest.raises(FileNotFoundError, match="epoch 3"):
        cli._models(cfg, "QWS", S, T, 3)
    with pytest.raises(FileNotFoundError, match="pretrained"):
        cli._models(cfg, "QWS", S, T, -1)
    net, low, high, k = cli._models(cfg, "QWS", S, T, 3, random_init=True)       # explicit: seeded random weights
    assert k == K
    sd_ml = oml.make_state_dict(128, 20, 2, 2, seed=7)
    sd_low, sd_high = opn.make_state_dict(H, 8), opn.make_state_dict(H, 9)
    (tmp_path / "solutions" / "ML" / "QWS").mkdir(parents=True)
    (tmp_path / "solutions" / "PNHigh" / "QWS").mkdir(parents=True)
    torch.save(sd_ml, "solutions/ML/QWS/model-3.pkl")                               # trainML.TrainML.start
    torch.save({"epoch": 4, "model": sd_high, "optimizer": {}}, "solutions/PNHigh/QWS/epoch3.model")      # trainPNHigh.TrainModel
    torch.save({"epoch": 4, "model": sd_low, "optimizer": {}}, "solutions/PNHigh/QWS/epoch3_low.model")
    net, low, high, _ = cli._models(cfg, "QWS", S, T, 3)
    assert all(torch.equal(v, sd_ml[k_]) for k_, v in net.state_dict().items())
    assert all(torch.equal(v, sd_low[k_]) for k_, v in low.state_dict().items())
    assert all(torch.equal(v, sd_high[k_]) for k_, v in high.state_dict().items())
    with pytest.raises(FileNotFoundError):                                          # another epoch: not these files
        cli._models(cfg, "QWS", S, T, 2)
    # the reference writes model-{n}.pkl as a pickle of the WHOLE module (trainML.py:147): a module object is taken by its
    # state_dict, anything that is neither is refused by name
    torch.save(net, "solutions/ML/QWS/model-3.pkl")                                 # a whole-module pickle of an importable class
    with pytest.raises(RuntimeError, match="GNNPN_TRUST_CHECKPOINT"):                # unpickling a module runs code: opt-in only
        cli._load_ml_checkpoint("solutions/ML/QWS/model-3.pkl")
    monkeypatch.setenv("GNNPN_TRUST_CHECKPOINT", "1")
    got = cli._load_ml_checkpoint("solutions/ML/QWS/model-3.pkl")
    monkeypatch.delenv("GNNPN_TRUST_CHECKPOINT")
    assert set(got) == set(sd_ml) and all(torch.equal(got[k_], sd_ml[k_]) for k_ in sd_ml)
    torch.save([1, 2, 3], "solutions/ML/QWS/model-3.pkl")
    with pytest.raises(RuntimeError, match="expected a state_dict of Net"):
        cli._models(cfg, "QWS", S, T, 3)
    (tmp_path / "solutions" / "ML" / "QWS" / "model-3.pkl").write_bytes(b"not a checkpoint")
    with pytest.raises(RuntimeError, match="not loadable here"):
        cli._models(cfg, "QWS", S, T, 3)


def test_calc_penalties():
    from gnnpn_sc_amd.ML2PN import calc
    qos = [[0.2, 0.4], [0.5, 0.3], [0.9, 0.9], [1.0, 1.0]]
    base = 0.5 * (0.3 + 1 - 0.3)
    assert abs(calc(qos, [[0.5, 1.0], [0.5, 1.0]]) - base) < 1e-12
    assert abs(calc(qos, [[0.9, 1.0], [0.5, 1.0]]) - (base + 1)) < 1e-12      # 0.81 < 0.9
    assert abs(calc(qos, [[0.9, 1.0], [0.1, 0.5]]) - (base + 2)) < 1e-12      # and 1.0 > 0.5


def test_state_dict_layout_is_the_reference_layout():
    from gnnpn_sc_amd.modelML import Net
    from gnnpn_sc_amd.modelPN import CombinatorialRL, reward
    net = Net(128, 300, 20, 2, 4)
    sd = oml.make_state_dict(128, 20, 2, 4, 0)
    assert set(net.state_dict()) == set(sd)
    assert all(net.state_dict()[k].shape == sd[k].shape for k in sd)
    assert net.state_dict()["serviceConvs.0.weight"].shape == (24, 256)        # in x out (PyG 1.7)
    pn = CombinatorialRL(0, 256, 235, 0, 10, 1, reward, "Dot", 5, 47)
    want = opn.make_state_dict(256, 0)
    assert set(pn.state_dict()) == set(want)
    assert "actor.alpha" not in pn.state_dict()                                # plain tensor in the reference
    ck = {"epoch": 3, "model": want, "optimizer": {}}                          # trainPNLow.py:112-117
    pn.load_state_dict(ck["model"])
    # 'Bahdanau': the Attention modules' parameters under the reference's names (modelPN.py:83-90,159-160)
    bah = CombinatorialRL(0, 32, 18, 2, 10, 1, reward, "Bahdanau", 3, 6)
    want_b = opn.make_state_dict(32, 0, attention="Bahdanau")
    assert set(bah.state_dict()) == set(want_b)
    assert all(bah.state_dict()[k].shape == want_b[k].shape for k in want_b)
    assert bah.state_dict()["actor.pointer.W_ref.weight"].shape == (32, 32, 1)
    bah.load_state_dict(want_b, strict=True)
    with pytest.raises(NotImplementedError):
        CombinatorialRL(0, 32, 18, 0, 10, 1, reward, "Luong", 3, 6)            # modelPN.py:116-117


def test_csr_layout_preserves_edge_order():
    from gnnpn_sc_amd import graph
    ei = torch.tensor([[3, 1, 2, 1, 0, 3], [0, 0, 1, 0, 2, 2]])
    w = torch.arange(6).float()
    csr = graph.csr_by_destination(ei, 4, w)
    assert csr.rowptr.tolist() == [0, 3, 4, 6, 6]
    assert csr.col.tolist() == [3, 1, 1, 2, 0, 3] and csr.w.tolist() == [0., 1., 3., 2., 4., 5.]
    g = graph.gcn_csr(torch.tensor([[0, 1, 1], [1, 1, 0]]), torch.tensor([0.5, 7.0, 0.25]), 3)
    # self loop of node 1 keeps weight 7; nodes 0 and 2 get weight-1 loops; loops come last in a row
    assert g.rowptr.tolist() == [0, 2, 4, 5]
    assert g.col.tolist() == [1, 0, 0, 1, 2] and g.w.tolist() == [0.25, 1.0, 0.5, 7.0, 1.0]
    with pytest.raises(ValueError):
        graph.segment_ptr(torch.tensor([0, 1, 0]), 2)


def test_pack_lstm_weight_layout():
    from gnnpn_sc_amd import ops
    H = 8
    w = torch.arange(4 * H * H, dtype=torch.float32).view(4 * H, H)
    p = ops.pack_lstm_weight(w)
    assert p.shape == (H // 4, 4, H, 4)
    assert float(p[1, 2, 5, 3]) == float(w[2 * H + 5, 1 * 4 + 3])


def test_synth_is_seeded_and_well_formed():
    import gnnpn_sc_amd.synth as synth
    a, b = synth.make_dataset(6, 40, 8, seed=3, tasks_per_problem=3), synth.make_dataset(6, 40, 8, seed=3, tasks_per_problem=3)
    assert a == b
    assert len(a["nodefeatures"][0][0]) == 6 + 1 + 6 and sum(a["labels"][0]) == 3
    t = synth.make_service_table(47, 2507, 0, degree=32)
    assert t.cat_ptr[-1] == 2507 and (np.diff(t.cat_ptr) >= 53).all()
    ei = t.edge_index
    assert (ei[0, 0::2] == ei[1, 1::2]).all() and (ei[1, 0::2] == ei[0, 1::2]).all()   # symmetric pairs


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from gnnpn_sc_amd import dist
    dist.init_process_group("gloo")
    B, T = 7, 5
    full = torch.arange(B * T, dtype=torch.int32).view(B, T)
    sizes = [dist.shard_range(B, r, world)[1] - dist.shard_range(B, r, world)[0] for r in range(world)]
    lo, hi = dist.shard_range(B, rank, world)
    ragged = dist.all_gather_indices(full[lo:hi], sizes)
    even = dist.all_gather_indices(full[rank * 3:(rank + 1) * 3])
    t = dist.max_over_ranks(1.0 + rank, torch.device("cpu"), world)
    dist.barrier(world)
    q.put((rank, torch.equal(ragged, full), torch.equal(even, full[:6]), t))
    dist.destroy(world)


def test_all_gather_of_indices_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True, True, 2.0), (1, True, True, 2.0)]


def _gloo8_worker(rank, world, port, q):
    """What a bench.py rank does around its steps at N = 8 (bench.py::main): environment as torchrun / self_launch set it,
    strong-scaling shard of the global batch, the asynchronous equal-shard all-gather of idx [B/N, T] -> [B, T], the
    barrier + MAX-over-ranks timing reduction."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world))
    from gnnpn_sc_amd import dist
    dist.init_process_group("gloo")
    G, T = 64, 47                                                     # global batch of a strong-scaling step, QWS steps
    full = (torch.arange(G * T, dtype=torch.int32) * 7 % 235).view(G, T)
    lo, hi = dist.shard_range(G, rank, world)
    out, work = dist.all_gather_indices_async(full[lo:hi].clone())
    if work is not None:
        work.wait()
    out2, _ = dist.all_gather_indices_async(full[lo:hi].clone(), out)     # the gathered buffer is reused step after step
    dist.barrier(world)
    t = dist.max_over_ranks(0.001 * (rank + 1), torch.device("cpu"), world)
    q.put((rank, tuple(out.shape), torch.equal(out, full), out2 is out or torch.equal(out2, full), round(t, 6)))
    dist.destroy(world)


def test_eight_rank_launch_path_gloo():
    """The N = 8 run belongs to the driver's 8-GPU node; its launch-side logic is rehearsed here with eight gloo ranks:
    rank environment, shard ranges that tile the global batch, all-gather shape [8 * B/8, T] in rank order, max-over-ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    world = 8
    procs = [ctx.Process(target=_gloo8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(r, (64, 47), True, True, 0.008) for r in range(world)]


def test_numa_binding_reads_sysfs_only(monkeypatch):
    """bench.bind_to_gpu_numa_node: no GPU in this container -> it reports why it did not bind, and it never raises."""
    import importlib
    import sys
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    bench = importlib.import_module("bench")
    info = bench.bind_to_gpu_numa_node(3)
    assert info["bound"] in (True, False) and ("why" in info or "numa_node" in info)
    assert bench.usable_cpus() >= 1


def test_batch_shard_is_a_partition():
    import gnnpn_sc_amd.synth as synth
    from gnnpn_sc_amd.pipeline import DeviceBatch
    table = synth.make_service_table(6, 40, 0, degree=4)
    pb = synth.make_problem_batch(table, 7, tasks_per_problem=3)
    full = DeviceBatch.from_problems(pb, torch.device("cpu"))
    parts = [full.shard(r, 3) for r in range(3)]
    assert sum(p.n_problems for p in parts) == 7
    assert torch.equal(torch.cat([p.x for p in parts]), full.x)
    assert torch.equal(torch.cat([p.present for p in parts]), full.present)
    for p in parts:
        assert int(p.seg_ptr[0]) == 0 and int(p.seg_ptr[-1]) == p.x.shape[0]
        assert int(p.wf_csr.rowptr[-1]) == p.wf_csr.col.numel()
        assert p.wf_csr.col.numel() == 0 or (int(p.wf_csr.col.min()) >= 0 and int(p.wf_csr.col.max()) < p.x.shape[0])


def _woa_driver_setup(tmp):
    """Rebuild the data set and the PNHigh actions file of tests/golden/woa_driver.json under ``tmp`` (cwd)."""
    import gnnpn_sc_amd.synth as synth
    import gnnpn_sc_amd.loadData as mine
    with open(os.path.join(GOLDEN, "woa_driver.json")) as f:
        fx = json.load(f)
    p = fx["params"]
    ds = synth.make_dataset(p["T"], p["S"], p["P"], seed=p["seed"], tasks_per_problem=p["tasks_per_problem"],
                            lo_range=tuple(p["lo_range"]))
    synth.write_dataset(tmp, "QWS", ds)
    os.makedirs(os.path.join(tmp, "solutions", "pretrained"))
    n_train = p["P"] // 4 * 3
    lists0, _, _ = mine.loadDataOther("QWS", False)
    actions = [[[0.0, 1.0, 1.0, 1.0, 0, 0, 0, 0] for _ in range(p["P"] - n_train)] for _ in range(p["T"])]
    for b, nodes in enumerate(ds["nodefeatures"][n_train:]):
        cats = [n[:-6].index(1) - 1 for n in nodes][1:]
        for l, c in enumerate(cats):
            k = fx["picks"][b][l]
            q = fx["foreign"][f"{b},{l}"] if k < 0 else list(lists0[b][l][k])
            actions[c][b] = q + [0, 0, 0, 0]
    with open(os.path.join(tmp, "solutions", "pretrained", "QWS-PNHigh.txt"), "w") as f:
        json.dump(actions, f)
    return fx, actions, n_train


def test_load_data_other_and_adds_match_reference(tmp_path, monkeypatch):
    """loadDataOther / addS (loadData.py:155-276) in both `reduct` settings of environment.ini against the digests of the
    reference's own output (tests/golden/make_golden_woa.py), and the oracle's qualities on a sample of the problems."""
    import hashlib
    import copy
    import gnnpn_sc_amd.loadData as mine
    from oracle import woa as owoa
    monkeypatch.chdir(tmp_path)
    fx, actions, n_train = _woa_driver_setup(str(tmp_path))
    p = fx["params"]
    n_test = p["P"] - n_train
    sols = [[actions[c][b][:4] for c in range(p["T"]) if sum(actions[c][b][:4]) != 3] for b in range(n_test)]
    ssets = [{tuple(round(v, 5) for v in r) for r in rows} for rows in sols]
    for reduct in (0, 0.55):
        want = fx["modes"][str(reduct)]
        lists, cons, mins = mine.loadDataOther("QWS", reduct, sSetList=ssets, train=False)
        assert hashlib.sha256(json.dumps([lists, cons]).encode()).hexdigest() == want["lists_sha256"]
        assert json.loads(json.dumps(lists[:3])) == want["lists_head"] and cons[:3] == want["constraints_head"]
        assert [sum(len(l) for l in q) for q in lists] == want["kept_per_problem"]
        for b in range(0, n_test, 9):                            # every 9th problem through the oracle
            r = owoa.eswoa(lists[b], cons[b], copy.deepcopy(sols[b]), p["popSize"], p["MAX_Iter"],
                           owoa.DrawStream(p["base_seed"] + n_train + b))
            assert mins[n_train + b] / r["best_fitness"] == want["quality"][b], (reduct, b)


def test_artifact_write_side_round_trips_through_check(fx, workdir):
    """The WRITE side of the artefact formats (ML2PN.write_artifacts, what ML2PN.infer / main.py --infer end with):
    rankings and actions produced by the CPU oracle chain are written in the reference's two formats, at the paths
    loadDataPN / check read (epoch -1 and epoch N), and check() scores them exactly as the oracle's check does."""
    from gnnpn_sc_amd import ML2PN
    from oracle import data as odata
    from oracle import pn as opn
    import torch
    ds, P, T, K = fx["dataset"], fx["P"], fx["T"], fx["K"]
    rankings = fx["rank_each"]
    rows, _ = odata.load_data_pn(ds["nodefeatures"], ds["serviceFeature"], rankings, ds["minCostList"], K)
    n_train = P // 4 * 3
    x = torch.tensor([odata.pn_inputs(r).tolist() for r in rows[n_train:]], dtype=torch.float32)
    out = opn.two_level_greedy(opn.make_state_dict(32, 1), opn.make_state_dict(32, 2), x, T, K)
    actions = [[out["actions"][j, t].double().tolist() for j in range(P - n_train)] for t in range(T)]
    k1, _ = odata.load_data_pn(ds["nodefeatures"], ds["serviceFeature"], rankings, ds["minCostList"], 1)
    want = odata.check(k1, ds["minCostList"], actions, T)
    for epoch in (-1, 7):
        p_rank, p_act = ML2PN.write_artifacts("QWS", epoch, rankings, actions)
        assert (p_rank, p_act) == ML2PN.artifact_paths("QWS", epoch) and os.path.exists(p_rank) and os.path.exists(p_act)
        with open(p_rank) as f:
            assert json.load(f) == rankings
        with open(p_act) as f:
            got = json.load(f)
        assert len(got) == T and len(got[0]) == P - n_train and len(got[0][0]) == 8
        if epoch != -1:      # check() always takes the rankings of epoch -1 (ML2PN.py:19) and the actions of `epoch`
            ML2PN.write_artifacts("QWS", -1, rankings, [[[0.0] * 8]])
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            res = ML2PN.check("QWS", T, epoch)
        assert abs(res - want) < 1e-12 and buf.getvalue().split()[0] == str(epoch)


def test_bench_strong_scaling_partition_is_the_same_global_set():
    """bench.py --scaling strong: the global batch is 8 seeded chunks; whatever the number of ranks, the union of the
    ranks' shards (in rank order) is the same problem set, so N=1 and N=8 work on identical data."""
    import bench
    import gnnpn_sc_amd.synth as synth
    w = dict(T=6, K=3, S=60, B=0, n_t=3, G=32)
    table = synth.make_service_table(w["T"], w["S"], seed=0, degree=4)
    whole = bench.rank_batches(synth, table, w, 0, 1, "strong", 2)
    for world in (2, 4, 8):
        for j in range(2):
            parts = [bench.rank_batches(synth, table, w, r, world, "strong", 2)[j] for r in range(world)]
            assert sum(p.n_problems for p in parts) == 32
            assert np.array_equal(np.concatenate([p.x for p in parts]), whole[j].x)
            assert np.array_equal(np.concatenate([p.local_bounds for p in parts]), whole[j].local_bounds)
            assert np.array_equal(np.concatenate([p.present for p in parts]), whole[j].present)
    assert not np.array_equal(whole[0].local_bounds, whole[1].local_bounds)       # resident batches differ
    b0 = whole[0]
    assert b0.batch.max() == 31 and b0.edge_index.max() == b0.x.shape[0] - 1       # offsets of the merged chunks


def test_half_batch_split_keeps_whole_tiles_in_the_first_half():
    """pipeline.half_batch_split: the cut the paired half-batches use (tiles of 16 problems never straddle it; both halves
    non-empty; the halves differ by at most one tile)."""
    from gnnpn_sc_amd.pipeline import half_batch_split
    assert [half_batch_split(b) for b in (0, 1, 16)] == [0, 0, 0]
    for b in list(range(17, 200)) + [256, 512, 520, 528, 1024, 4096, 4097]:
        h = half_batch_split(b)
        assert 0 < h < b and h % 16 == 0, (b, h)
        assert abs(h - (b - h)) <= 16 + 15, (b, h)
        assert -(-h // 16) + -(-(b - h) // 16) == -(-b // 16), (b, h)     # no extra tile


def test_run_checked_repeats_a_batch_once_after_a_failed_handoff(monkeypatch):
    """ops.run_checked: the batch is run again (attempt 1: the drivers switch to the write-through hand-off) when check_status
    reports a failed hand-off, with a warning; a second failure is raised; a clean run is not repeated."""
    import warnings
    from gnnpn_sc_amd import ops
    calls, fail = [], {"n": 1}

    def fake_check(device=None):
        if fail["n"] > 0:
            fail["n"] -= 1
            raise ops.GnnpnError("status 0x1 — an inter-workgroup hand-off timed out")
    monkeypatch.setattr(ops, "check_status", fake_check)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert ops.run_checked(lambda attempt: calls.append(attempt) or ("result", attempt)) == ("result", 1)
    assert calls == [0, 1] and len(w) == 1 and "failed hand-off" in str(w[0].message)
    calls.clear()
    assert ops.run_checked(lambda attempt: calls.append(attempt) or 7) == 7 and calls == [0]
    fail["n"] = 2
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(ops.GnnpnError):
            ops.run_checked(lambda attempt: None)


def test_embedding_runs_keep_their_artefacts_apart(tmp_path, monkeypatch):
    """embeddingTag=1 runs write and read under ./solutions/PN{Low,High}/<ds>/20embeddings/ (the reference appends "20embeddings/" to
    self.dataset after loading the candidate rows: trainPNHigh.py:197-198,237-240, trainPNLow.py:190-191), so they neither overwrite
    the embeddingTag=0 checkpoints nor pick them up.  Host logic only: the drivers' path functions, and PNHigh.start /
    PNLow.start up to the point where the paths are used (loadDataPN and TrainModel stubbed)."""
    from gnnpn_sc_amd import trainPNHigh as tp
    assert tp.artefact_dataset("QWS/", 0) == "QWS/" and tp.artefact_dataset("QWS/", 1) == "QWS/20embeddings/"
    assert tp.low_checkpoint_path("QWS/", 7) == "./solutions/PNLow/QWS//epoch7.model"
    assert tp.low_checkpoint_path("QWS/20embeddings/", 7) == "./solutions/PNLow/QWS/20embeddings//epoch7.model"
    assert tp.low_checkpoint_path("QWS/", -1) == "./solutions/pretrained/QWS-PNLow.model"
    assert tp.low_checkpoint_path("QWS/20embeddings/", -1) == "./solutions/pretrained/QWS/20embeddings-PNLow.model"
    monkeypatch.chdir(tmp_path)
    seen = {}

    class FakeTM:
        def __init__(self, model, train_ds, val_ds, epochDiv, beta, use_cuda, dataset, *a, **k):
            seen["tm_dataset"] = dataset

        def train_and_validate(self, *a):
            pass

    rows = [[[0.0] * 9 for _ in range(6)] for _ in range(8)]
    import gnnpn_sc_amd.loadData as ld
    monkeypatch.setattr(ld, "loadDataPN", lambda **k: (seen.setdefault("load_dataset", k["dataset"]) and None) or (rows, [[0] * 2] * 8))
    monkeypatch.setattr(tp, "TrainModel", FakeTM)
    monkeypatch.setattr(tp.evalPN, "SCDataset", lambda *a: None)

    class FakeNet:
        def load_state_dict(self, sd):
            seen["loaded"] = sd

        def to(self, dev):
            return self
    monkeypatch.setattr(tp, "_pointer_model", lambda *a, **k: FakeNet())
    import torch
    monkeypatch.setattr(torch, "load", lambda path, **k: seen.setdefault("low_path", path) and {"model": "sd"})
    for tag, want_dir, want_low in ((0, "QWS", "./solutions/PNLow/QWS//epoch3.model"),
                                    (1, "QWS/20embeddings", "./solutions/PNLow/QWS/20embeddings//epoch3.model")):
        seen.clear()
        tp.PNHigh("QWS", tag, False, 2, 1, 3, 32, 0, 10, True, 0.9, 2.0, 1e-4, 5, 3).start(1, device="cpu")
        assert seen["load_dataset"] == "QWS"                     # the candidate rows always come from the plain dataset directory
        assert seen["tm_dataset"] == want_dir and seen["low_path"] == want_low
        seen.clear()
        tp.PNLow("QWS", tag, False, 2, 1, 3, 32, 0, 10, True, 0.9, 2.0, 1e-4, 5).start(1, device="cpu")
        assert seen["load_dataset"] == "QWS" and seen["tm_dataset"] == want_dir
    p = tp.PNHigh("QWS", 1, False, 2, 1, 3, 32, 0, 10, True, 0.9, 2.0, 1e-4, 5, 3)
    p.start(1, device="cpu")
    p.start(1, device="cpu")                                      # a second start does not append twice (the reference would)
    assert seen["tm_dataset"] == "QWS/20embeddings"


def test_oracle_and_product_candidate_reduction_agree():
    """The whole-pipeline GPU tests take the oracle-side candidate rows from oracle.data.reduce_candidates (tests/parity.py::
    oracle_candidate_rows: the synthetic batch put into the reference's formats) — on the CPU the product's host helper of the
    artefact path, loadData.reduce_from_ranking, is held to it on random rankings (dummy rows for absent categories included)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import numpy as np
    import torch
    import gnnpn_sc_amd.synth as synth
    from gnnpn_sc_amd.loadData import reduce_from_ranking
    from parity import oracle_candidate_rows
    for T, S, K, n_t in ((12, 300, 3, 5), (6, 90, 5, 6), (20, 400, 2, 7)):
        table = synth.make_service_table(T, S, seed=T, degree=8)
        pb = synth.make_problem_batch(table, 5, seed=S, tasks_per_problem=n_t, lo_range=(0.2, 0.99))
        rng = np.random.default_rng(K)
        rank = [rng.permutation(S) for _ in range(5)]
        want = oracle_candidate_rows(pb, table, rank, K, min(T, n_t) + 1)
        cat_of = np.repeat(np.arange(T), np.diff(table.cat_ptr))
        got = torch.tensor([reduce_from_ranking(rank[i], pb.local_bounds[i], pb.present[i], pb.global_bounds[i], cat_of, table.qos, K)
                            for i in range(5)], dtype=torch.float32)[:, :, 1:]
        assert torch.equal(got, want), (T, S, K)


def test_partitioned_device_is_announced_once(monkeypatch):
    """A device that shows fewer than 256 compute units (CPX / DPX / QPX partition) gets ONE RuntimeWarning per device from the
    pipeline constructor, naming what will raise and what will run slowly; a full device gets none."""
    import warnings
    import torch
    from gnnpn_sc_amd import pipeline as pl

    class Props:
        def __init__(self, n):
            self.multi_processor_count = n
    pl._warned_partitioned.clear()
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: Props(64 if i == 1 else 256))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        pl.warn_partitioned_device(torch.device("cuda", 0), "split")
        pl.warn_partitioned_device(torch.device("cuda", 1), "split")
        pl.warn_partitioned_device(torch.device("cuda", 1), "split")
        pl.warn_partitioned_device(torch.device("cpu"), "split")
    assert len(w) == 1 and "64 compute units" in str(w[0].message) and "GNNPN_E_UNSUP" in str(w[0].message)
    pl._warned_partitioned.clear()


def test_slot_streams_keep_normal_priority_without_an_rccl_process_group():
    """pipeline._has_collective_stream: only an RCCL ("nccl") process group brings a stream of its own that the slots' streams must
    not share a hardware queue with (profiles/LOG_r06.md 9a); no process group — and gloo, whose collectives run on the host — do not."""
    from gnnpn_sc_amd import pipeline
    import torch.distributed as td
    assert not td.is_initialized() and pipeline._has_collective_stream() is False
