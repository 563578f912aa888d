"""-m gpu: the GNN mirror (Net) against the golden vectors of the reference's own Net glue, and the
fused device pipeline (scores -> candidates -> two-level decode -> reward) against the oracle chain."""
import os
import numpy as np
import pytest
import torch

from conftest import eager_reference, golden
from oracle import data as odata
from oracle import ml as oml
from oracle import pn as opn
from parity import R_ATOL, assert_index_parity, robust_problems

pytestmark = pytest.mark.gpu
SCORE_ATOL = 1e-5


def make_net(fx, dev, vocab=100):
    from gnnpn_sc_amd.modelML import Net
    net = Net(int(fx["hidden"]), int(fx["S"]), int(fx["emb"]), int(fx["n_gin"]), int(fx["n_gcn"]), vocab=vocab,
              isServices=bool(fx["is_services"]) if "is_services" in fx else True)
    net.load_state_dict(oml.make_state_dict(int(fx["hidden"]), int(fx["emb"]), int(fx["n_gin"]), int(fx["n_gcn"]),
                                            int(fx["seed"]) + 2, vocab=vocab), strict=True)
    return net.to(dev).eval()


@pytest.mark.parametrize("name", ["tiny", "qws", "normal", "noservices"])
def test_net_forward_golden(dev, name):
    fx = golden(f"ml_{name}.npz")
    net = make_net(fx, dev)
    B, S = int(fx["B"]), int(fx["S"])
    t = lambda k: torch.from_numpy(fx[k]).to(dev)   # noqa: E731
    # feed what a PyG batch holds: B replicated copies of the service graph (trainML.py:109-114)
    data = oml.make_data(t("x"), t("edge_index"), t("batch"), t("x_service").repeat(B, 1),
                         torch.cat([t("edge_index_service") + b * S for b in range(B)], 1),
                         t("edge_attr_service").repeat(B))
    scores = net(data)
    assert scores.shape == (B, S) and scores.dtype == torch.float32
    err = float((scores.cpu() - torch.from_numpy(fx["scores"])).abs().max())
    assert err < SCORE_ATOL, err
    # ranking: identical wherever the golden scores are separated by more than the tolerance
    from gnnpn_sc_amd import ops
    rank = ops.rank_rows(scores).cpu().long()
    want = torch.from_numpy(fx["ranking"])
    gs = torch.from_numpy(fx["scores"])
    sorted_scores = torch.gather(gs, 1, want)
    gap_ok = torch.ones_like(want, dtype=torch.bool)
    gap = (sorted_scores[:, :-1] - sorted_scores[:, 1:]) > 4 * SCORE_ATOL
    gap_ok[:, 1:] &= gap
    gap_ok[:, :-1] &= gap
    assert torch.equal(rank[gap_ok], want[gap_ok])
    # second forward reuses the cached CSR and is deterministic
    assert torch.equal(net(data), scores)


def test_net_forward_on_a_pyg_style_batch(dev):
    """Net.forward must do what the reference's forward does on the batch it is GIVEN.  ml_pygbatch.npz is the
    reference's own glue on two graphs batched as torch_geometric 1.7.0 batches them (copy 1's service edges shifted by
    graph 0's workflow node count): the GCN runs over both copies with those edges and the copies are averaged."""
    fx = golden("ml_pygbatch.npz")
    net = make_net(fx, dev)
    B, S = int(fx["B"]), int(fx["S"])
    t = lambda k: torch.from_numpy(fx[k]).to(dev)   # noqa: E731
    ei, ea = oml.pyg_batch_service_edges(t("edge_index_service"), t("edge_attr_service"), fx["offsets"])
    data = oml.make_data(t("x"), t("edge_index"), t("batch"), t("x_service").repeat(B, 1), ei, ea)
    scores = net(data)
    err = float((scores.cpu() - torch.from_numpy(fx["scores"])).abs().max())
    assert err < SCORE_ATOL, err
    one = oml.make_data(t("x"), t("edge_index"), t("batch"), t("x_service"), t("edge_index_service"), t("edge_attr_service"))
    single = net(one)
    assert float((single.cpu() - torch.from_numpy(fx["scores_single_copy"])).abs().max()) < SCORE_ATOL
    assert float((single - scores).abs().max()) > 1e-3           # the two semantics really differ on this batch
    bad = oml.make_data(t("x"), t("edge_index"), t("batch"), t("x_service").repeat(3, 1), ei, ea)
    with pytest.raises(ValueError):
        net(bad)                                                  # neither one copy nor one copy per graph


def test_net_rejects_training(dev):
    fx = golden("ml_tiny.npz")
    net = make_net(fx, dev).train()
    with pytest.raises(NotImplementedError):
        net(oml.make_data(*[torch.from_numpy(fx[k]).to(dev) for k in
                            ("x", "edge_index", "batch", "x_service", "edge_index_service", "edge_attr_service")]))


def _oracle_chain(table, pb, sd_ml, sd_low, sd_high, n_gin, n_gcn, K):
    data = oml.make_data(torch.from_numpy(pb.x), torch.from_numpy(pb.edge_index), torch.from_numpy(pb.batch),
                         torch.from_numpy(table.x_service), torch.from_numpy(table.edge_index),
                         torch.from_numpy(table.edge_attr))
    scores = oml.net_forward(sd_ml, data, n_gin, n_gcn)
    return scores


@pytest.mark.parametrize("T,S,K,B,n_t,H", [(6, 60, 3, 8, 3, 32), (47, 940, 5, 24, 10, 256)])
def test_pipeline_vs_oracle_chain(dev, T, S, K, B, n_t, H):
    import gnnpn_sc_amd.synth as synth
    from gnnpn_sc_amd.modelML import Net
    from gnnpn_sc_amd.modelPN import CombinatorialRL, reward
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline
    table = synth.make_service_table(T, S, seed=5, degree=8)
    pb = synth.make_problem_batch(table, B, seed=6, tasks_per_problem=n_t, lo_range=(0.85, 0.96))
    sd_ml = oml.make_state_dict(128, 20, 2, 2, seed=7)
    sd_low, sd_high = opn.make_state_dict(H, 8), opn.make_state_dict(H, 9)
    net = Net(128, S, 20, 2, 2)
    net.load_state_dict(sd_ml)
    low = CombinatorialRL(0, H, T * K, 0, 10, 1, reward, "Dot", K, T, level="Low")
    high = CombinatorialRL(0, H, T * K, 0, 10, 1, reward, "Dot", K, T, level="High")
    low.load_state_dict(sd_low)
    high.load_state_dict(sd_high)
    pipe = ML2PNPipeline(net.to(dev).eval(), low.to(dev).eval(), high.to(dev).eval(), K)
    out = pipe.run(DeviceServices.from_table(table, dev), DeviceBatch.from_problems(pb, dev))

    # stage 1: scores
    ref_scores = _oracle_chain(table, pb, sd_ml, sd_low, sd_high, 2, 2, K)
    assert float((out["scores"].cpu() - ref_scores).abs().max()) < SCORE_ATOL
    # stage 2: candidate reduction — exact given the device's own scores (stable ranking rule)
    ds_like = {str(c + 1): table.qos[table.cat_ptr[c]:table.cat_ptr[c + 1]].tolist() for c in range(T)}
    rank = oml.rank_services(out["scores"].cpu()).tolist()
    nodes_per = pb.x.shape[0] // B
    rows_want = []
    for b in range(B):
        nodes = []
        for r in pb.x[b * nodes_per:(b + 1) * nodes_per]:
            onehot = [0] * (T + 1)
            onehot[int(r[0])] = 1
            nodes.append(onehot + [float(v) for v in r[1:].astype(np.float64)])
        rows_want.append(odata.reduce_candidates(rank[b], nodes, ds_like, K)[0])
    rows_want = torch.tensor(rows_want, dtype=torch.float32)[:, :, 1:]
    assert torch.equal(out["pn_inputs"].cpu(), rows_want)
    # and it is the same reduction the oracle's own scores give, wherever scores are well separated
    # stage 3: decode given identical inputs
    ref = opn.two_level_greedy(sd_low, sd_high, rows_want, T, K)
    robust = robust_problems(ref["margin_low"], ref["margin_high"])
    s = assert_index_parity(out["idx_low"], ref["idx_low"], robust, "pipe/low", 0.8, rows_want) & \
        assert_index_parity(out["idx_high"], ref["idx_high"], robust, "pipe/high", 0.8, rows_want)
    assert bool(robust.any())
    assert float((out["R"].cpu()[s] - ref["R"][s]).abs().max()) <= R_ATOL
    # "selected service indices" (north star): the chosen service ids agree with the oracle's
    ids = out["candidate_ids"].cpu().long()
    sel_got = torch.gather(ids, 1, out["idx_high"].cpu().long())
    sel_ref = torch.gather(ids, 1, ref["idx_high"])
    assert torch.equal(sel_got[s], sel_ref[s])
    # sharding the batch does not change any result (what dist.py relies on)
    svc = DeviceServices.from_table(table, dev)
    full = DeviceBatch.from_problems(pb, dev)
    parts = [pipe.run(svc, full.shard(r, 2)) for r in range(2)]
    assert torch.equal(torch.cat([p["idx_high"] for p in parts]), out["idx_high"])
    assert torch.equal(torch.cat([p["R"] for p in parts]), out["R"])


def test_full_size_properties(dev):
    """BASELINE configs[1] size (QWS shape, B=256): size-independent properties instead of an
    oracle run — every pick lies in its step's window, picks of present categories are feasible
    candidates, dummy rows only in absent categories, rewards are finite and rounded to 5 decimals,
    the run is deterministic, and a batch permutation permutes the outputs."""
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline
    T, S, K, B = 47, 2507, 5, 256
    table = synth.make_service_table(T, S, seed=0, degree=32)
    pb = synth.make_problem_batch(table, B, seed=1, tasks_per_problem=10)
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K)
    svc, batch = DeviceServices.from_table(table, dev), DeviceBatch.from_problems(pb, dev)
    out = pipe.run(svc, batch)
    idx = out["idx_high"].cpu().long()
    step = torch.arange(T).unsqueeze(0)
    assert bool(((idx >= step * K) & (idx < (step + 1) * K)).all())
    ids = out["candidate_ids"].cpu().view(B, T, K)
    present = torch.from_numpy(pb.present).bool()
    assert bool((ids[~present] == -1).all()) and bool((ids[present] >= 0).all())
    cat_of = torch.repeat_interleave(torch.arange(T), torch.from_numpy(np.diff(table.cat_ptr)).long())
    real = ids >= 0
    assert torch.equal(cat_of[ids[real].long()], torch.arange(T).view(1, T, 1).expand(B, T, K)[real])
    R = out["R"].cpu().double()
    assert bool(torch.isfinite(R).all()) and float((R * 1e5 - (R * 1e5).round()).abs().max()) < 2e-2
    out2 = pipe.run(svc, batch)
    assert torch.equal(out2["idx_high"], out["idx_high"]) and torch.equal(out2["R"], out["R"])
    # reversing the batch reverses the outputs
    rev = DeviceBatch.from_problems(synth.ProblemBatch(
        pb.x.reshape(B, -1, 7)[::-1].reshape(-1, 7).copy(), pb.edge_index, pb.batch, pb.local_bounds[::-1].copy(),
        pb.present[::-1].copy(), pb.global_bounds[::-1].copy()), dev)
    out3 = pipe.run(svc, rev)
    assert torch.equal(out3["idx_high"].flip(0), out["idx_high"]) and torch.equal(out3["R"].flip(0), out["R"])


def test_trainml_test_and_eval_block_mirrors(dev):
    """TrainML.test (rankings + P@1/P@5) and the PNHigh eval block (SCDataset + allActions) on a dataset in
    the reference's JSON format, against the oracle."""
    import json, os
    from conftest import GOLDEN
    from gnnpn_sc_amd import loadData as ld
    from gnnpn_sc_amd.evalPN import SCDataset, evaluate
    from gnnpn_sc_amd.modelML import Net
    from gnnpn_sc_amd.modelPN import CombinatorialRL, reward
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline
    fx = json.load(open(os.path.join(GOLDEN, "data_small.json")))
    ds, K, T, S, P = fx["dataset"], fx["K"], fx["T"], fx["S"], fx["P"]
    table, pb = ld.tables_from_dataset(ds)
    sd_ml = oml.make_state_dict(128, 20, 2, 2, seed=3)
    net = Net(128, S, 20, 2, 2)
    net.load_state_dict(sd_ml)
    H = 32
    low = CombinatorialRL(0, H, T * K, 0, 10, 1, reward, "Dot", K, T, level="Low")
    high = CombinatorialRL(0, H, T * K, 0, 10, 1, reward, "Dot", K, T, level="High")
    low.load_state_dict(opn.make_state_dict(H, 1))
    high.load_state_dict(opn.make_state_dict(H, 2))
    pipe = ML2PNPipeline(net.to(dev).eval(), low.to(dev).eval(), high.to(dev).eval(), K)
    labels = torch.tensor(ds["labels"], dtype=torch.float32)
    ranking, (p1, p5) = pipe.test(DeviceServices.from_table(table, dev), DeviceBatch.from_problems(pb, dev),
                                  labels.to(dev))
    want_p = oml.precision_at(ranking.cpu().long(), labels)
    assert abs(p1 - want_p[0]) < 1e-6 and abs(p5 - want_p[1]) < 1e-6
    assert sorted(ranking[0].tolist()) == list(range(S))
    # eval block on the reference-format rows of the shared-ranking reduction
    val = SCDataset(fx["rows_shared"], ds["minCostList"])
    assert len(val) == P and val[0][0].shape == (T * K, 8) and val[0][1] == ds["minCostList"][0]
    acts, tour = evaluate(low, high, val, T, batch_size=5, device=str(dev))
    assert len(acts) == T and len(acts[0]) == P and len(acts[0][0]) == 8 and len(tour) == 4
    x = torch.stack([val[i][0] for i in range(P)])
    ref = opn.two_level_greedy(opn.make_state_dict(H, 1), opn.make_state_dict(H, 2), x, T, K)
    robust = robust_problems(ref["margin_low"], ref["margin_high"])
    got = torch.tensor(acts).permute(1, 0, 2)                      # [P,T,8]
    same = (got == ref["actions"]).all(-1).all(-1)
    assert bool(same[robust].all()) and bool(robust.any())


def test_pipelined_runner_matches_single_stream(dev):
    """PipelinedRunner (2 captured graphs on 2 streams, private workspaces, new batches copied into the
    static inputs) gives, batch by batch, what a plain single-stream run of the same kernels gives."""
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd import ops
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    T, S, K, B = 47, 940, 5, 48
    table = synth.make_service_table(T, S, seed=0, degree=16)
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K)
    svc = DeviceServices.from_table(table, dev)
    batches = [DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=10 + i, tasks_per_problem=10), dev)
               for i in range(5)]
    runner = PipelinedRunner(pipe, svc, batches[0], slots=2)
    got = []
    for b in batches:
        out, slot = runner.submit(b)
        with torch.cuda.stream(runner.stream(slot)):          # consume before the slot is reused
            got.append((out["idx_high"].clone(), out["R"].clone(), out["candidate_ids"].clone()))
    runner.synchronize()
    ops.check_status(dev)
    for b, (idx, R, ids) in zip(batches, got):
        ref = eager_reference(pipe, svc, b, decode_impl=runner.decode_impl)                  # the decoder form the 2-slot runner uses
        assert torch.equal(ids, ref["candidate_ids"]) and torch.equal(idx, ref["idx_high"]) and torch.equal(R, ref["R"])
    with pytest.raises(ops.GnnpnError):
        runner.submit(DeviceBatch.from_problems(synth.make_problem_batch(table, B + 1, seed=3, tasks_per_problem=10), dev))


def test_runner_counts_its_work_and_degrades_itself(dev):
    """Proof of work through graph replays, and the runner's own degraded mode (round 5): every replay books the workgroup-tiles
    its cooperative launches are expected to finish (device) and the host books the same per replay (ops.graph_replay); poll()
    finds finished == expected == host count.  A status raised in any slot — here planted by hand: a failure code in one slot, a
    missing tile in the other — makes poll() return it and switches the runner to the write-through hand-off for good (new
    graphs over the same static buffers); the batches submitted afterwards are right."""
    import warnings
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd import ops
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    T, S, K, B = 47, 940, 5, 48
    table = synth.make_service_table(T, S, seed=0, degree=16)
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K)
    svc = DeviceServices.from_table(table, dev)
    batches = [DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=70 + i, tasks_per_problem=10), dev) for i in range(4)]
    runner = PipelinedRunner(pipe, svc, batches[0], slots=2, write_through=False)
    assert runner.poll() == 0
    for b in batches:
        runner.submit(b)
    assert runner.poll() == 0
    units = ops.Workspaces.coop_units(2, B)                       # two nets, three tiles of 16 problems, eight members
    for p in runner.progress():                                   # per slot: the capture's two warm-up passes + two replays
        for part in ("encoder", "decoder"):
            assert p[part]["finished"] == p[part]["expected"] == p[part]["host_expected"] == 4 * units, p
    # slot 0: a failure code as a kernel would raise it; slot 1: one workgroup-tile short
    runner.workspaces[0].status[0] = 2
    runner.workspaces[1].status[7] -= 1
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        word = runner.poll()
    assert word == (2 | ops.Workspaces.SHORTFALL) and runner.write_through and runner.degraded == word
    assert any("write-through" in str(w.message) for w in caught)
    got = []
    for b in batches:
        out, slot = runner.submit(b)
        with torch.cuda.stream(runner.stream(slot)):
            got.append((out["idx_high"].clone(), out["R"].clone()))
    assert runner.poll() == 0                                     # counters restarted after the failure, and agree again
    for b, (idx, R) in zip(batches, got):
        ref = eager_reference(pipe, svc, b, decode_impl=runner.decode_impl)
        assert torch.equal(idx, ref["idx_high"]) and torch.equal(R, ref["R"])
    runner.workspaces[0].status[0] = 1                            # synchronize(check=True) raises instead
    with pytest.raises(ops.GnnpnError, match="status 0x1"):
        runner.synchronize()


def test_first_replays_of_a_burst_start_together(dev):
    """(The common start is OPT-IN since round 5 — GNNPN_PIPE_COMMON_START_US — the slots no longer slip apart once the front
    half shares the encoders' LDS footprint: test_front_half_shares_the_cooperative_footprint.)
    Two free-running slots: the first replay after the runner was waited for (synchronize / poll, or the first ever) and the
    other slot's first one are held behind ONE gate (gnnpn_gate_wait on slot 0's transfer stream, opened by the host when the second
    replay is enqueued, by synchronize / poll, or after pipeline.COMMON_START_US) — the later
    ones are not, batches from host memory are not, a runner without the option is not; the results are what a single stream
    gives.  (What the common start is worth is a measurement: tools/probes/stagger_probe.py, profiles/LOG_r01_r04.md section 13.4.)"""
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd import ops, pipeline
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    T, S, K, B = 47, 940, 5, 48
    table = synth.make_service_table(T, S, seed=0, degree=16)
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K)
    svc = DeviceServices.from_table(table, dev)
    batches = [DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=40 + i, tasks_per_problem=10), dev) for i in range(3)]
    assert pipeline.COMMON_START_US == 0 and PipelinedRunner(pipe, svc, batches[0], slots=2).common_start_us == 0    # the default: off
    os.environ["GNNPN_PIPE_COMMON_START_US"] = "400"
    try:
        runner = PipelinedRunner(pipe, svc, batches[0], slots=2)
    finally:
        del os.environ["GNNPN_PIPE_COMMON_START_US"]
    assert runner.common_start_us == 400 and runner._drained and runner._gate is None

    def burst(n, source):
        starts, outs = [], []
        for i in range(n):
            s = runner.count % runner.n_slots
            out, slot = runner.submit(source[i % len(source)])
            assert slot == s
            starts.append(runner._gate is not None)
            with torch.cuda.stream(runner.stream(slot)):
                outs.append((i % len(source), out["idx_high"].clone(), out["R"].clone()))
        return starts, outs

    starts, outs = burst(5, batches)
    assert starts == [True, False, False, False, False]          # (the gate is pending after the first held replay, gone after the second)
    assert not runner._drained
    runner.synchronize()
    assert runner._drained
    starts2, outs2 = burst(4, batches)
    assert starts2[0] and not any(starts2[1:])
    runner.synchronize()
    ops.check_status(dev)
    for j, idx, R in outs + outs2:
        ref = eager_reference(pipe, svc, batches[j], decode_impl=runner.decode_impl)
        assert torch.equal(idx, ref["idx_high"]) and torch.equal(R, ref["R"])
    # batches from pinned host memory: no hold (their transfers share the copy engine; slots in step wait for each other's)
    host = [runner.pack_host(b) for b in batches[:2]]
    runner.synchronize()
    runner.submit(host[0])
    assert runner._gate is None and not runner._drained
    runner.submit(host[1])
    runner.synchronize()
    ops.check_status(dev)
    # the option off (the default)
    plain = PipelinedRunner(pipe, svc, batches[0], slots=2)
    assert plain.common_start_us == 0
    plain.submit(batches[0])
    assert plain._gate is None
    plain.synchronize()


def test_front_half_shares_the_cooperative_footprint(dev):
    """Round 5: the ordinary LDS-using kernels in front of the encoder (one-launch GIN branch, score product) are captured with
    the cooperative kernels' LDS footprint in a runner of two free-running slots in the exact-split precision (78 KB: a freed
    range then fits the other slot's encoder workgroup — what made the slots slip apart, profiles/r05_slip_front_lds.jsonl);
    the setting is the calling thread's and is restored; f32 runners and the half-batch mode pad nothing; results unchanged."""
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd import _lib, ops
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    T, S, K, B = 47, 940, 5, 48
    table = synth.make_service_table(T, S, seed=0, degree=16)
    net, low, high = build_models(T, S, K, dev)
    svc = DeviceServices.from_table(table, dev)
    batches = [DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=90 + i, tasks_per_problem=10), dev) for i in range(3)]
    lib = _lib.load()
    assert lib.gnnpn_lds_footprint_kb(-1) == 0                        # (out of range: a query)
    with ops.lds_footprint(78):
        assert lib.gnnpn_lds_footprint_kb(-1) == 78
        with ops.lds_footprint(0):
            assert lib.gnnpn_lds_footprint_kb(-1) == 0
        assert lib.gnnpn_lds_footprint_kb(-1) == 78
    assert lib.gnnpn_lds_footprint_kb(-1) == 0
    pipe = ML2PNPipeline(net, low, high, K)                           # exact split by default
    runner = PipelinedRunner(pipe, svc, batches[0], slots=2)
    assert runner.front_lds_kb == 78 == runner.lds_kb[0] and lib.gnnpn_lds_footprint_kb(-1) == 0
    assert PipelinedRunner(ML2PNPipeline(net, low, high, K, precision="f32"), svc, batches[0], slots=2).front_lds_kb == 0
    assert PipelinedRunner(pipe, svc, batches[0], slots=2, halves=True).front_lds_kb == 0
    got = []
    for b in batches:
        out, slot = runner.submit(b)
        with torch.cuda.stream(runner.stream(slot)):
            got.append((out["idx_high"].clone(), out["R"].clone(), out["scores"].clone()))
    runner.synchronize()
    for b, (idx, R, sc) in zip(batches, got):
        ref = eager_reference(pipe, svc, b, decode_impl=runner.decode_impl)          # eager: no padding
        assert torch.equal(idx, ref["idx_high"]) and torch.equal(R, ref["R"]) and torch.equal(sc, ref["scores"])


@pytest.mark.parametrize("B,precision", [(88, "f32"), (88, "split"), (50, "split"), (33, "f32")])
def test_half_batches_side_by_side_match_the_whole_batch(dev, B, precision):
    """The runner's form for batches of 512 problems and more — the recurrent part as two half-batches on two streams,
    private workspaces, joined before the reward — forced on at small sizes (whole tiles, a ragged last tile, a second half
    of one ragged tile): graph replays give, batch by batch, the bits of the plain one-stream run of the whole batch."""
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd import ops
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    T, S, K = 47, 940, 5
    table = synth.make_service_table(T, S, seed=0, degree=16)
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K, precision=precision)
    svc = DeviceServices.from_table(table, dev)
    batches = [DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=20 + i, tasks_per_problem=10), dev)
               for i in range(3)]
    runner = PipelinedRunner(pipe, svc, batches[0], slots=2, halves=True)
    assert runner.halves and runner.n_slots == 2 and runner.n_streams == 1 and len(runner.workspaces) == 2
    keys = ("idx_low", "idx_high", "R", "actions", "action_probs", "win_low", "win_high_raw", "candidate_ids")
    got = []
    for b in batches:
        out, slot = runner.submit(b)
        with torch.cuda.stream(runner.stream(slot)):
            got.append({k: out[k].clone() for k in keys})
    runner.synchronize(check=True)
    ops.check_status(dev)
    for b, g in zip(batches, got):
        ref = eager_reference(pipe, svc, b, decode_impl=runner.decode_impl)
        for k in keys:
            assert g[k].shape == ref[k].shape and torch.equal(g[k], ref[k]), k
    # the documented output contract holds in this mode too: a slot's static outputs stay intact until the slot
    # comes round again, n_slots submits later — read here, unconsumed, after the NEXT submission has run
    out0, s0 = runner.submit(batches[1])
    out1, s1 = runner.submit(batches[2])
    assert s0 != s1 and out0["idx_high"].data_ptr() != out1["idx_high"].data_ptr()
    runner.synchronize(check=True)
    for out, b in ((out0, batches[1]), (out1, batches[2])):
        ref = eager_reference(pipe, svc, b, decode_impl=runner.decode_impl)
        for k in keys:
            assert torch.equal(out[k], ref[k]), k
    with pytest.raises(ops.GnnpnError):                     # nothing to put on the second stream
        pipe.run(svc, DeviceBatch.from_problems(synth.make_problem_batch(table, 16, seed=3, tasks_per_problem=10), dev),
                 ws=(ops.new_workspaces(dev), ops.new_workspaces(dev)))


def test_all_gather_indices_rccl_world1(dev):
    """The one collective of the path on its real backend (backend "nccl" is RCCL on ROCm), world size 1: the equal-shard
    and the ragged form of dist.all_gather_indices on device tensors.  (World 2 runs on CPU with gloo, test_host_logic.)"""
    import os
    import socket
    import torch.distributed as td
    from gnnpn_sc_amd import dist as gdist
    if td.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    # round 6 (profiles/LOG_r06.md 9a): beside a process group's stream the slots' streams take a priority — hardware queues — of their own
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd import ops
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    T, S, K, B = 47, 940, 5, 48
    table = synth.make_service_table(T, S, seed=0, degree=16)
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K)
    svc = DeviceServices.from_table(table, dev)
    batches = [DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=20 + i, tasks_per_problem=10), dev) for i in range(4)]
    early = PipelinedRunner(pipe, svc, batches[0], slots=2)       # created BEFORE the process group: normal priority ...
    assert early.stream_priority == 0
    try:
        rank, world = gdist.init_process_group("nccl", dev)
        assert (rank, world) == (0, 1)
        idx = torch.arange(4 * 47, dtype=torch.int32, device=dev).view(4, 47)
        assert torch.equal(gdist.all_gather_indices(idx), idx)
        assert torch.equal(gdist.all_gather_indices(idx, sizes=[4]), idx)
        assert gdist.max_over_ranks(1.5, dev, 2) == 1.5          # the all-reduce(MAX) bench.py uses, forced through RCCL
        with pytest.warns(RuntimeWarning, match="before the RCCL process group"):    # ... and it says so at its first submit
            early.submit(batches[0])
        early.synchronize()
        runner = PipelinedRunner(pipe, svc, batches[0], slots=2)
        assert runner.stream_priority == -1 and PipelinedRunner(pipe, svc, batches[0], slots=2, stream_priority=0).stream_priority == 0
        got, gathers = [], {}
        for b in batches:
            out, slot = runner.submit(b)
            with torch.cuda.stream(runner.stream(slot)):          # the path's collective behind the step, on the group's own stream
                gathers[slot] = gdist.all_gather_indices_async(out["idx_high"], gathers.get(slot, (None, None))[0])
                if gathers[slot][1] is not None:
                    gathers[slot][1].wait()
                got.append((gathers[slot][0].clone(), out["R"].clone()))
        runner.synchronize()
        ops.check_status(dev)
        for b, (gi, R) in zip(batches, got):
            ref = eager_reference(pipe, svc, b, decode_impl=runner.decode_impl)
            assert torch.equal(gi.view_as(ref["idx_high"]), ref["idx_high"]) and torch.equal(R, ref["R"])
    finally:
        if td.is_initialized():
            td.destroy_process_group()


def test_infer_writes_artifacts_that_check_scores(dev, tmp_path, monkeypatch):
    """ML2PN.infer (the write side of the artefact formats) on a dataset in the reference's JSON formats: the rankings
    and actions files it writes have the reference's shapes, `check` reads them back, and they equal the CPU oracle
    chain's (rankings wherever the oracle's scores are separated, actions on problems whose decisions are robust);
    `python main.py QWS ML+2PN -1 --infer` does the same from the command line and prints the reference's line."""
    import contextlib
    import io
    import json
    import os
    import subprocess
    import sys
    import gnnpn_sc_amd.synth as synth
    from conftest import ROOT
    from gnnpn_sc_amd import ML2PN
    from gnnpn_sc_amd.modelML import Net
    from gnnpn_sc_amd.modelPN import CombinatorialRL, reward
    T, S, P, K, H = 6, 60, 16, 3, 256
    ds = synth.make_dataset(T, S, P, seed=3, tasks_per_problem=3, lo_range=(0.80, 0.955))
    synth.write_dataset(str(tmp_path), "QWS", ds)
    monkeypatch.chdir(tmp_path)
    sd_ml = oml.make_state_dict(128, 20, 2, 2, seed=7)
    sd_low, sd_high = opn.make_state_dict(H, 8), opn.make_state_dict(H, 9)
    net = Net(128, S, 20, 2, 2)
    net.load_state_dict(sd_ml)
    low = CombinatorialRL(0, H, T * K, 0, 10, 1, reward, "Dot", K, T, level="Low")
    high = CombinatorialRL(0, H, T * K, 0, 10, 1, reward, "Dot", K, T, level="High")
    low.load_state_dict(sd_low)
    high.load_state_dict(sd_high)
    p_rank, p_act = ML2PN.infer("QWS", net, low, high, K, epoch=-1, device=str(dev), batch_size=5)
    with open(p_rank) as f:
        rankings = json.load(f)
    with open(p_act) as f:
        actions = json.load(f)
    n_test = P // 4
    assert len(rankings) == P and sorted(rankings[0]) == list(range(S))
    assert len(actions) == T and len(actions[0]) == n_test and len(actions[0][0]) == 8
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        got = ML2PN.check("QWS", T, -1)
    assert buf.getvalue().split()[0] == "-1" and np.isfinite(got)
    # the oracle chain on the same files
    import gnnpn_sc_amd.loadData as ld
    table, pb = ld.tables_from_dataset(ds, 0, P)
    data = oml.make_data(torch.from_numpy(pb.x), torch.from_numpy(pb.edge_index), torch.from_numpy(pb.batch),
                         torch.from_numpy(table.x_service), torch.from_numpy(table.edge_index), torch.from_numpy(table.edge_attr))
    scores = oml.net_forward(sd_ml, data, 2, 2)
    want_rank = oml.rank_services(scores)
    ss = torch.gather(scores, 1, want_rank)
    sep = torch.ones_like(want_rank, dtype=torch.bool)
    gap = (ss[:, :-1] - ss[:, 1:]) > 4 * SCORE_ATOL
    sep[:, 1:] &= gap
    sep[:, :-1] &= gap
    assert torch.equal(torch.tensor(rankings)[sep], want_rank[sep])
    rows, _ = odata.load_data_pn(ds["nodefeatures"], ds["serviceFeature"], rankings, ds["minCostList"], K)
    x = torch.tensor([odata.pn_inputs(r).tolist() for r in rows[P - n_test:]], dtype=torch.float32)
    ref = opn.two_level_greedy(sd_low, sd_high, x, T, K)
    robust = robust_problems(ref["margin_low"], ref["margin_high"])
    got_act = torch.tensor(actions).permute(1, 0, 2).float()                      # [n_test, T, 8]
    same = (got_act == ref["actions"]).all(-1).all(-1)
    assert bool(same[robust].all()) and bool(robust.any())
    if bool(same.all()):
        k1, _ = odata.load_data_pn(ds["nodefeatures"], ds["serviceFeature"], rankings, ds["minCostList"], 1)
        assert abs(odata.check(k1, ds["minCostList"], actions, T) - got) < 1e-12
    # the command line: weights from ./solutions/pretrained/ in the reference's checkpoint formats
    os.remove(p_rank), os.remove(p_act)
    torch.save(sd_ml, "solutions/pretrained/QWS-ML.pt")
    torch.save({"epoch": 0, "model": sd_low, "optimizer": {}}, "solutions/pretrained/QWS-PNLow.model")
    torch.save({"epoch": 0, "model": sd_high, "optimizer": {}}, "solutions/pretrained/QWS-PNHigh.model")
    with open("environment.ini", "w") as f:
        f.write("[QWS-ML]\nnumLayersGIN = 2\nnumLayersGCN = 2\nhiddenChannels = 128\nembeddingChannels = 20\ndropout = 0.0\n"
                f"[QWS-PNHigh]\nserNumber = {K}\nhidden_size = {H}\nn_glimpses = 0\ntanh_exploration = 10\nuse_tanh = 1\n"
                f"[QWS-ML+2PN]\nserviceCategory = {T}\nepoch = -1\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "QWS", "ML+2PN", "-1", "--infer"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1].split()
    assert line[0] == "-1" and abs(float(line[1]) - got) < 1e-9                   # same weights, same data -> same score
    assert os.path.exists(p_rank) and os.path.exists(p_act)


def test_paired_slots_with_batches_from_pinned_host_memory(dev, monkeypatch):
    """Slots that start in pairs (``lockstep``; forced here at the QWS shape) fed from pinned host arenas: ``submit`` holds the
    leader back until its partner is submitted and enqueues both replays behind both transfers; ``after`` runs on the slot's
    stream right behind the replay (the results' way home); an odd submission is released by ``synchronize`` / ``stream``.  Every
    step's results equal the single-stream run's, in submission order."""
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    monkeypatch.setenv("GNNPN_PIPE_LOCKSTEP", "1")
    T, S, K, B = 47, 940, 5, 256
    table = synth.make_service_table(T, S, seed=0, degree=16)
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K)
    svc = DeviceServices.from_table(table, dev)
    batches = [DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=70 + i, tasks_per_problem=10), dev) for i in range(3)]
    runner = PipelinedRunner(pipe, svc, batches[0], slots=2)
    assert runner.lockstep and not runner.halves
    refs = [eager_reference(pipe, svc, b, decode_impl=runner.decode_impl) for b in batches]
    hosts = [runner.pack_host(b) for b in batches]
    got, order = [], []

    def home(out, s):                                        # on the slot's stream, behind its replay
        h = {k: torch.empty_like(out[k], device="cpu").pin_memory() for k in ("idx_high", "R")}
        for k in h:
            h[k].copy_(out[k], non_blocking=True)
        got.append(h)
        order.append(s)
    n = 7                                                    # odd: the last leader has no partner
    for i in range(n):
        runner.submit(hosts[i % 3], after=home)
        assert len(got) == (i + 1) // 2 * 2                  # a leader's replay (and its `after`) waits for the partner's submission
    assert runner._deferred is not None
    runner.synchronize(check=True)                           # releases the waiting leader
    assert runner._deferred is None and len(got) == n and order == [i % 2 for i in range(n)]
    for i, h in enumerate(got):
        assert torch.equal(h["idx_high"], refs[i % 3]["idx_high"].cpu()) and torch.equal(h["R"], refs[i % 3]["R"].cpu()), i
    runner.submit(hosts[0], after=home)                      # a waiting leader is also released by asking for its stream
    assert len(got) == n
    runner.stream(1).synchronize()
    assert len(got) == n + 1
    runner.synchronize(check=True)
    assert torch.equal(got[-1]["idx_high"], refs[0]["idx_high"].cpu())


def test_two_slots_whichever_starts_first(dev):
    """Placement by claim (csrc/coop_common.h): every cooperative launch staffs all its groups one workgroup per CU no
    matter what else is resident.  Round 1's LDS-footprint steering only held while slot 0 led; here the two-slot runner
    is started with slot 1 first, then with slot 0 first, on an idle chip: every launch reports all its members placed
    (ops.Workspaces.placement), no hand-off failure, and the results equal the single-stream run."""
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd import ops
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    T, S, K, B = 47, 940, 5, 256
    table = synth.make_service_table(T, S, seed=0, degree=16)
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K, precision="f32")
    svc = DeviceServices.from_table(table, dev)
    batch = DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=4, tasks_per_problem=10), dev)
    runner = PipelinedRunner(pipe, svc, batch, slots=2)
    assert runner.lds_kb == [0, 0]                               # no footprint steering (fp32 builds: 19 / 27 KB, nothing to equalise)
    assert PipelinedRunner(ML2PNPipeline(net, low, high, K), svc, batch, slots=2).lds_kb == [78, 78]   # the default (exact split): ONE footprint
    ref = eager_reference(pipe, svc, batch, decode_impl=runner.decode_impl)
    for first in (1, 0, 1):
        torch.cuda.synchronize()
        runner.count = first
        for _ in range(6):
            runner.submit()
        runner.synchronize(check=True)
        for s in range(2):
            o = runner.graphs[s].outputs
            assert torch.equal(o["idx_high"], ref["idx_high"]) and torch.equal(o["R"], ref["R"])
            w = runner.workspaces[s]
            pl = w.placement()                                   # the last encoder launch's own counters (per-XCD words summed)
            assert pl["members_placed"] == 256, pl               # 32 groups x 8 members, one per CU
            assert pl["off_canonical_seats"] <= 2, pl            # every member on its CU's canonical seat (a reserve take-over is rare: 0-2 per 500 steps)


@pytest.mark.parametrize("delay_us", [200, 400, 535])
def test_two_slots_out_of_phase(dev, delay_us):
    """Two free-running slots that do NOT start together (round 6, profiles/LOG_r06.md section 16).  Slot 1's first replay of every round
    is held back by a spin kernel, so the slots run out of phase for the whole round and their cooperative launches start beside the
    other slot's ORDINARY kernels (small LDS ranges at the bottom of a CU) and, as the phases drift, beside each other.  The first
    one-round-trip placement of this round counted a launch into the per-device "staffing" count up to eight times for a microsecond;
    every workgroup that read the count then took another launch to be staffing, badly placed ones declined their seats until only the
    reserve was left, seats went off their canonical CUs 2 ms later, and a third of such rounds ended in a hand-off time-out (0.3 s
    each).  Here: every round status 0, (next to) no seat off its CU, results equal to the single-stream run, no round slower than 1.5 x the median."""
    import statistics
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    T, S, K, B = 47, 2507, 5, 256
    table = synth.make_service_table(T, S, seed=0, degree=32)
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K)
    svc = DeviceServices.from_table(table, dev)
    batch = DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=4, tasks_per_problem=T), dev)
    runner = PipelinedRunner(pipe, svc, batch, slots=2, auto_degrade=False)
    ref = eager_reference(pipe, svc, batch, decode_impl=runner.decode_impl)
    for _ in range(8):
        runner.submit()
    runner.synchronize(check=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(10_000_000)
    e1.record()
    torch.cuda.synchronize()
    cyc_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)
    ms, words, off = [], [], 0
    for _ in range(12):
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for s_ in range(2):
            runner.stream(s_).wait_event(t0)
        with torch.cuda.stream(runner.stream(1)):
            torch.cuda._sleep(int(delay_us * cyc_per_us))
        for _ in range(60):
            runner.submit()
        cur = torch.cuda.current_stream()
        for s_ in range(2):
            cur.wait_stream(runner.stream(s_))
        t1.record()
        torch.cuda.synchronize()
        ms.append(t0.elapsed_time(t1))
        off += sum((w.last_seats or {}).get("off_canonical", 0) for w in runner.workspaces)
        words.append(runner.poll())
        for s_ in range(2):
            o = runner.graphs[s_].outputs
            assert torch.equal(o["idx_high"], ref["idx_high"]) and torch.equal(o["R"], ref["R"])
    assert not any(words), [hex(v) for v in words]
    med = statistics.median(ms)
    assert max(ms) < 1.5 * med, ms
    seats = [w.last_seats for w in runner.workspaces]
    # seats off their canonical CU: the reserve taking over after 2 ms is legitimate and rare (0-2 per 500 steps in ordinary bursts,
    # tools/probes/dbg_offcanonical.py); the regression produced 7-10 per LAUNCH (hundreds over these 720 steps)
    assert all((st or {}).get("off_canonical", 0) <= 16 for st in seats), seats
    from conftest import record_agreement
    record_agreement(f"two_slots_out_of_phase_{delay_us}us", {"rounds": len(ms), "steps_per_round": 60, "round_ms_median": round(med, 3),
                                                           "round_ms_max": round(max(ms), 3), "status_words": words, "seats": seats})


def test_more_than_two_slots_still_two_in_flight(dev):
    """``slots`` > 2 = more static input / output sets, NOT more replays in flight: a CU holds two cooperative workgroups, and a third
    free-running stream ended in half-staffed launches and time-outs (round 6, tools/probes/dbg_three_slots.py: status 0x13, 24 k problems/s).
    The slots are dealt onto two streams in turn; every slot's outputs equal the single-stream run, status 0."""
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    T, S, K, B = 47, 940, 5, 256
    table = synth.make_service_table(T, S, seed=0, degree=16)
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K)
    svc = DeviceServices.from_table(table, dev)
    batches = [DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=30 + i, tasks_per_problem=10), dev) for i in range(4)]
    for slots in (3, 4):
        runner = PipelinedRunner(pipe, svc, batches[0], slots=slots, auto_degrade=False)
        assert runner.n_slots == slots and runner.n_streams == 2 and not runner.lockstep
        refs = [eager_reference(pipe, svc, b, decode_impl=runner.decode_impl) for b in batches]
        got = []
        for i in range(24):
            out, s_ = runner.submit(batches[i % 4])
            with torch.cuda.stream(runner.stream(s_)):
                got.append((i % 4, out["idx_high"].clone(), out["R"].clone()))
        assert runner.poll() == 0
        for j, idx, R in got:
            assert torch.equal(idx, refs[j]["idx_high"]) and torch.equal(R, refs[j]["R"])


def test_two_runners_take_turns(dev):
    """Two PipelinedRunners of one process, fed alternately (round 6, tools/probes/dbg_two_runners.py): four cooperative launches in front
    of the two workgroup slots of every CU used to end in half-staffed launches and bounded-wait time-outs (status 0x13, 18 ms per step).
    PipelinedRunner._take_turn makes a runner's first submit after another runner's wait, stream-side, for what that runner has
    enqueued: every launch finds the slots it was built for.  Status 0, no seat off its CU, results equal to the single-stream runs."""
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    T, S, K, B = 47, 940, 5, 256
    table = synth.make_service_table(T, S, seed=0, degree=16)
    svc = DeviceServices.from_table(table, dev)
    runners, refs = [], []
    for seed in (0, 1):
        net, low, high = build_models(T, S, K, dev, seed=seed)
        pipe = ML2PNPipeline(net, low, high, K)
        batch = DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=5 + seed, tasks_per_problem=10), dev)
        r = PipelinedRunner(pipe, svc, batch, slots=2, auto_degrade=False)
        runners.append(r)
        refs.append(eager_reference(pipe, svc, batch, decode_impl=r.decode_impl))
    for pattern in ([0, 1] * 40, [0, 0, 0, 1, 1, 0, 1, 1, 1, 0] * 8):
        for i in pattern:
            runners[i].submit()
        words = [r.poll() for r in runners]
        assert words == [0, 0], [hex(v) for v in words]
        for r, ref in zip(runners, refs):
            assert all((w.last_seats or {}).get("off_canonical", 0) <= 16 for w in r.workspaces), [w.last_seats for w in r.workspaces]   # (rare, legitimate; without turn-taking: hundreds)
            for s_ in range(2):
                o = r.graphs[s_].outputs
                assert torch.equal(o["idx_high"], ref["idx_high"]) and torch.equal(o["R"], ref["R"])


def test_two_slots_beside_long_ordinary_kernels(dev):
    """The staffing reserve of coop_place (csrc/coop_common.h).  At 1000 tasks x 5000 candidates x 512 problems the other
    slot's front-end kernels fill whole CUs for longer than the early surplus workgroups of a cooperative launch last; before
    the reserve existed about half of the bench runs at this size ended with decoder groups one or two members short
    (tools/repro_synth4.sh).  24 pipelined steps: no failed hand-off, results equal to the single-stream run."""
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    T, S, K, B = 1000, 5000, 5, 512
    table = synth.make_service_table(T, S, seed=0, degree=32)
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K)
    svc = DeviceServices.from_table(table, dev)
    batch = DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=4, tasks_per_problem=T), dev)
    runner = PipelinedRunner(pipe, svc, batch, slots=2, halves=False)      # two whole batches in flight (the default at this size pairs half-batches)
    assert runner.n_slots == 2 and runner.lockstep                       # 5000 recurrent steps: pairs start together
    ref = eager_reference(pipe, svc, batch, decode_impl=runner.decode_impl)
    for _ in range(24):
        runner.submit()
    runner.synchronize(check=True)
    for s in range(2):
        o = runner.graphs[s].outputs
        assert torch.equal(o["idx_high"], ref["idx_high"]) and torch.equal(o["R"], ref["R"])


def test_two_slots_started_together_with_long_recurrences(dev):
    """256 problems x 5000 recurrent steps, exact split (78 KB of LDS per cooperative workgroup), two slots whose replays
    start together while the host consumes every result in between.  Before coop_place looked at WHERE in the CU's LDS a
    workgroup had been put, a workgroup that landed above a short-lived neighbour of the other slot's front half kept its
    78 KB in the middle of the CU, the partner launch's workgroup for that CU could not be placed, both launches stayed
    under-staffed and one step in ten ended in a hand-off time-out (wrong picks, status 0x1).  60 steps: every result equal
    to the single-stream run, no status raised."""
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    T, S, K, B = 1000, 5000, 5, 256
    table = synth.make_service_table(T, S, seed=0, degree=32)
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K, precision="split")
    svc = DeviceServices.from_table(table, dev)
    batches = [DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=300 + i, tasks_per_problem=T), dev) for i in range(3)]
    runner = PipelinedRunner(pipe, svc, batches[0], slots=2)
    assert runner.n_slots == 2 and runner.lockstep and not runner.halves
    refs = [eager_reference(pipe, svc, b, decode_impl=runner.decode_impl) for b in batches]
    torch.cuda.synchronize()      # the runner's streams do not wait for this one: a third cooperative launch in flight is not a supported mix
    pending, bad = [], []
    for i in range(60):
        out, s = runner.submit(batches[i % 3])
        ev = torch.cuda.Event()
        ev.record(runner.stream(s))
        pending.append((i, ev, out))
        if len(pending) == 2:
            j, evj, oj = pending.pop(0)
            evj.synchronize()
            if not all(torch.equal(oj[k], refs[j % 3][k]) for k in ("idx_low", "idx_high", "R")):
                bad.append(j)
    runner.synchronize(check=True)
    from gnnpn_sc_amd import _lib, ops
    ops.check_status(dev)
    assert not bad, bad
    assert _lib.load().gnnpn_coop_staffing_count() == 0     # every launch left the count of launches that are staffing


@pytest.mark.parametrize("precision", ["f32", "split"])
def test_soak_two_slots(dev, precision):
    """2,000 pipelined steps, a different batch every step, eager work on the default stream in between (what recycled the
    runtime's staging memory under the captured memset nodes this used to rely on): every output of every step equals
    the single-stream run of the same batch, bit for bit, and no launch reports a failed hand-off.
    (tools/soak_pipeline.py is the 20,000-step form.)"""
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    T, S, K, B = 47, 940, 5, 64
    table = synth.make_service_table(T, S, seed=0, degree=16)
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K, precision=precision)
    svc = DeviceServices.from_table(table, dev)
    batches = [DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=200 + i, tasks_per_problem=10), dev)
               for i in range(6)]
    runner = PipelinedRunner(pipe, svc, batches[0], slots=2)
    refs = [eager_reference(pipe, svc, b, decode_impl=runner.decode_impl) for b in batches]
    keys = ("idx_low", "idx_high", "R", "actions")
    pending, bad = [], 0
    for i in range(2000):
        v = (i * 5 + i // 7) % len(batches)
        out, s = runner.submit(batches[v])
        ev = torch.cuda.Event()
        ev.record(runner.stream(s))
        pending.append((v, ev, out))
        if len(pending) == 2:
            vj, evj, oj = pending.pop(0)
            evj.synchronize()
            bad += int(not all(torch.equal(oj[k], refs[vj][k]) for k in keys))
    for vj, evj, oj in pending:
        evj.synchronize()
        bad += int(not all(torch.equal(oj[k], refs[vj][k]) for k in keys))
    runner.synchronize(check=True)
    assert bad == 0, f"{bad} of 2000 pipelined steps differ from the single-stream run"
    from gnnpn_sc_amd import _lib
    assert _lib.load().gnnpn_coop_staffing_count() == 0     # 8,000 cooperative launches later nobody is left in the count of staffing launches


@pytest.mark.parametrize("shape", ["qws_two_slots", "synth4_half_batches"])
def test_soak_beside_a_collective_shaped_interferer(dev, shape):
    """What an 8-rank RCCL ring kernel does to the placement of the cooperative launches cannot be seen on a one-GPU box (RCCL
    at world size 1 is a copy), so a stand-in runs beside the pipeline: every step, on a third stream, 8-32 workgroups that
    hold 64-96 KB of LDS each for 20-50 us (gnnpn_debug_lds_interferer), at random phases — beside two free-running slots at
    the QWS shape and beside the paired half-batches of a 512-problem batch at the 1000-task shape.  No launch reports a
    failed hand-off and every step's outputs equal the single-stream run's; the throughput lost is measured with a burst beside
    EVERY step and beside every 8th (the cadence of bench.py's bucketed all-gather) and recorded (profiles/r04_interferer_soak.json)."""
    import json
    import os
    import random
    import time
    import gnnpn_sc_amd.synth as synth
    from bench import build_models
    from gnnpn_sc_amd import _lib
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
    T, S, K, B, n_t, steps = (47, 2507, 5, 256, 10, 600) if shape == "qws_two_slots" else (1000, 5000, 5, 512, 1000, 24)
    table = synth.make_service_table(T, S, seed=0, degree=32)
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K)
    svc = DeviceServices.from_table(table, dev)
    batches = [DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=300 + i, tasks_per_problem=n_t), dev) for i in range(2)]
    runner = PipelinedRunner(pipe, svc, batches[0], slots=2)
    assert runner.halves == (shape != "qws_two_slots")
    packed = [runner.pack(b) for b in batches]
    # the single-stream reference runs use the device's default workspaces: conftest.eager_reference checks their status like the
    # runner's and repeats a failed attempt once (seen on some boxes of the pool; profiles/LOG_r01_r04.md section 13.3); repeats are recorded
    import warnings
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        refs = [eager_reference(pipe, svc, b, decode_impl=runner.decode_impl) for b in batches]
    eager_retries = sum("failed hand-off" in str(c.message) for c in caught)
    keys = ("idx_low", "idx_high", "R")
    side = torch.cuda.Stream()
    lib = _lib.load()
    rng = random.Random(7)

    def run(every):                                                           # every: a burst every that many steps (0: none)
        pending, bad = [], 0
        for i in range(8):
            runner.submit(packed[i % 2])
        runner.synchronize(check=True)
        t0 = time.perf_counter()
        for i in range(steps):
            out, s = runner.submit(packed[i % 2])
            if every and i % every == 0:
                for _ in range(rng.randint(1, 2)):                            # one or two bursts, on their own stream
                    _lib.check(lib.gnnpn_debug_lds_interferer(rng.randint(8, 32), rng.choice((64, 80, 96)) * 1024, rng.randint(20, 50),
                                                              side.cuda_stream), "gnnpn_debug_lds_interferer")
            ev = torch.cuda.Event()
            ev.record(runner.stream(s))
            pending.append((i % 2, ev, out))
            if len(pending) == runner.n_slots:
                vj, evj, oj = pending.pop(0)
                evj.synchronize()
                bad += int(not all(torch.equal(oj[k], refs[vj][k]) for k in keys))
        for vj, evj, oj in pending:
            evj.synchronize()
            bad += int(not all(torch.equal(oj[k], refs[vj][k]) for k in keys))
        runner.synchronize(check=True)                                        # raises on any failed hand-off
        side.synchronize()
        return B * steps / (time.perf_counter() - t0), bad

    run(0)                                                                    # clocks, first launches
    base, bad0 = run(0)
    with_it, bad1 = run(1)                                                    # a burst beside EVERY step
    with_8, bad3 = run(8)                                                     # beside every 8th: the cadence of bench.py's bucketed all-gather at this step length
    base2, bad2 = run(0)
    ref_rate = 0.5 * (base + base2)
    rec = {"shape": shape, "steps": steps, "problems_per_s_without": [round(base, 1), round(base2, 1)],
           "problems_per_s_with_interferer_every_step": round(with_it, 1), "loss_pct_every_step": round((1 - with_it / ref_rate) * 100, 2),
           "problems_per_s_with_interferer_every_8th_step": round(with_8, 1), "loss_pct_every_8th_step": round((1 - with_8 / ref_rate) * 100, 2),
           "interferer": "1-2 bursts of 8-32 workgroups x 64-96 KB LDS x 20-50 us on a third stream",
           "placement_last_launch": [w.placement() for w in runner.workspaces],
           "steps_with_wrong_outputs": {"without": [bad0, bad2], "every_step": bad1, "every_8th_step": bad3},
           "reference_runs_repeated_after_a_failed_handoff": eager_retries}
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity", f"interferer_soak_{shape}.json"), "w") as f:
        json.dump(rec, f)
    assert bad0 == bad1 == bad2 == bad3 == 0, rec
    assert lib.gnnpn_coop_staffing_count() == 0
    # measured at the QWS shape: 5-17 % / -1.5-6 % over round 5's runs (600 steps are ~1 s: the rates themselves move by a few %).  Round 6:
    # beside an interferer at EVERY step the rate is set by the interferer, not by the launches — 322-394 k on this round's tree, 331-397 k on
    # round 5's, same box, alternating (tools/r06/ab_interferer.sh) — so against a base rate that rose from 424 k to 443 k the same absolute
    # rate reads as a larger loss (11-32 % against 6-22 %): the bound on the every-step case is 0.60 now; every 8th step (the cadence of
    # bench.py's bucketed all-gather) stays at 0.88
    assert with_it >= 0.60 * ref_rate and with_8 >= 0.88 * ref_rate, rec


def test_bench_line_contract(dev):
    """bench.py as the driver runs it (a child process, `--gpus 1 --steps K --warmup W`): exit 0, exactly ONE JSON line on
    stdout with the contract's keys, the BASELINE metric / config, a roofline object for the dominant kernel and the CPU
    baseline; and the self-launching `--gpus 4` entry (four ranks on this one GPU over gloo: launch path only), weak and strong."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()                 # an earlier test's process group left RANK / WORLD_SIZE behind
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT", "GNNPN_FORCE_DIST")}
    env["PYTHONPATH"] = root
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--min-time", "0",
                        "--no-split-line"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert base["metric"].startswith(d["metric"].split(" (")[0]) and d["unit"] == "problems/s"
    assert d["dtype"].split()[0] == "f32-split3xf16" and d["precision"] == "split"     # the default arithmetic has its own dtype token ...
    assert d["agreement_vs_f32"]["identical_decisions"] >= 0.999                     # ... and always reports its agreement with plain f32
    assert d["vs_baseline"] is None and d["value"] > 0                               # BASELINE.md publishes no number
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0 < rf["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert cb["processes"] >= 1 and cb["configurations"] and cb["usable_cpus"] >= cb["cores"] >= 1
    pr = d["per_rank"]
    assert len(pr) == 1 and pr[0]["rank"] == 0 and pr[0]["degraded"] is None and "degraded" not in d
    assert all(p["members_placed"] == 256 for p in pr[0]["placement_last_launch"])            # 32 groups x 8 members per slot
    # round 6: placement events summed over EVERY launch of the run (sticky status words 1-2, ABI 9) — none on an idle chip — with the
    # booked encoder tiles they are summed over; no process group at N = 1, so no collective latency; and the committed counter
    # summaries are either quoted as "current" or named as stale (never silently quoted)
    assert pr[0]["seats"]["declined"] <= 8 and pr[0]["seats"]["off_canonical"] <= 8 and pr[0]["seats"]["encoder_tiles_booked"] > 0   # (0 as a rule; a reserve take-over is rare and legitimate)
    assert pr[0]["collective_ms"] is None and d["config"]["slot_stream_priority"] == 0         # normal-priority slot streams without a process group
    files = rf["counters_from"]["files"]
    assert files and all(v == "current" or v.startswith("stale") for v in files.values())
    assert (rf["traffic"] is not None) == all(v == "current" for k, v in files.items() if "pmc_traffic" in k and "aggregate" not in k)
    assert d["timing"]["untimed_settling_rounds"] == 0                                        # --min-time 0: the contract's bare region
    # a hand-off status during the timed rounds (forced) does not end the run: this rank switches, in process, to the degraded
    # form — one step in flight, write-through hand-off, the collective (RCCL, world 1) waited for before the next launch — the
    # phase is repeated and the line says so
    env_d = dict(env, GNNPN_BENCH_FORCE_DEGRADE="timed", GNNPN_FORCE_DIST="1")
    rd = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--min-time", "0.05",
                         "--no-split-line", "--no-cpu-baseline", "--no-kernel-timers"], capture_output=True, text=True, timeout=600,
                        env=env_d, cwd=root)
    assert rd.returncode == 0, rd.stderr[-2000:]
    dd = json.loads([ln for ln in rd.stdout.splitlines() if ln.strip()][-1])
    assert dd["degraded"][0]["phase"] == "timed rounds" and dd["degraded"][0]["status"] == 0x40 and dd["per_rank"][0]["degraded"]
    assert dd["value"] > 0
    cm = dd["per_rank"][0]["collective_ms"]                                                   # the process group's first and second all-gather, host clock
    assert cm["first"] > 0 and cm["second"] > 0
    assert dd["config"]["slot_stream_priority"] == 0      # the DEGRADED form's: one step in flight on one stream keeps normal priority (the two-slot
    #                                                        runner beside RCCL takes -1: test_all_gather_indices_rccl_world1; LOG_r06 9a)
    # the self-launching multi-rank entry, FOUR ranks on this one GPU over gloo (launch path only: rank environment, port,
    # rank-0-only stdout, NUMA binding, all-gather shape [N * B, T]; the box allows six GPU processes, pytest is one of them;
    # the N = 8 logic is rehearsed with eight gloo ranks on the CPU in tests/test_host_logic.py) — weak, then strong scaling
    env2 = dict(env, GNNPN_BENCH_SHARE_GPU="1")
    for extra, gb in ((["--batch", "32"], 4 * 32), (["--scaling", "strong", "--global-batch", "128"], 128)):
        r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1", "--min-time", "0",
                             "--no-cpu-baseline", "--no-split-line", "--no-kernel-timers"] + extra, capture_output=True, text=True,
                            timeout=900, env=env2, cwd=root)
        assert r2.returncode == 0, r2.stderr[-2000:]
        out2 = [ln for ln in r2.stdout.splitlines() if ln.strip()]
        assert len(out2) == 1, out2
        d2 = json.loads(out2[0])
        assert d2["n_gpus"] == 4 and d2["config"]["global_batch"] == gb == 4 * d2["config"]["batch_per_gpu"]
        assert "rank0_cpu_affinity" in d2["config"] and "NOT_A_MEASUREMENT" in d2["config"]
        assert [p["rank"] for p in d2["per_rank"]] == [0, 1, 2, 3] and all(p["ms_per_step_local"] > 0 for p in d2["per_rank"])
    # ... and with ONE of the four ranks reporting a hand-off status after the warm-up: that rank degrades, every rank repeats
    # the phase (the collectives of a phase are counted the same on all ranks), the line names the rank
    r3 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1", "--min-time", "0",
                         "--no-cpu-baseline", "--no-split-line", "--no-kernel-timers", "--batch", "32"], capture_output=True, text=True,
                        timeout=900, env=dict(env2, GNNPN_BENCH_FORCE_DEGRADE="warmup", GNNPN_BENCH_FORCE_DEGRADE_RANK="2"), cwd=root)
    assert r3.returncode == 0, r3.stderr[-2000:]
    d3 = json.loads([ln for ln in r3.stdout.splitlines() if ln.strip()][-1])
    assert [x["rank"] for x in d3["degraded"]] == [2] and [bool(p["degraded"]) for p in d3["per_rank"]] == [False, False, True, False]
