"""-m gpu: parity at BASELINE.json's sizes and shapes against fixtures produced by the REAL reference
(tests/golden/make_golden_big.py -> pn_big_*.npz): configs[1] QWS B=256 (two batches), configs[2] Normal B=1024,
configs[3] T=1000/K=5 (L=5000), configs[4] T=2000/K=10 (L=20000, also with the fp16-operand encoder) — run on the
builds bench.py times (decode_impl 4, LDS-footprint placement, HIP-graph replay on two streams, f32 and split), plus
full-batch properties and the whole pipeline (vocab > 100) against the live oracle at reduced B.

Every test records its measured agreement (conftest.record_agreement -> gpurun_out/parity/*.json); the merged record
of THIS round is committed as tests/golden/agreement_r06.json (tools/collect_parity.py; agreement_r02 / r03 / r05.json are the
earlier rounds') and the floors asserted here are the measured values."""
import numpy as np
import pytest
import torch

from conftest import eager_reference, golden, record_agreement
from oracle import ml as oml
from oracle import pn as opn
from parity import LOGIT_ATOL, assert_R_parity, oracle_candidate_rows, prefix_parity
from pn_inputs import pn_inputs_chunked

pytestmark = pytest.mark.gpu


def _nets(fx, dev):
    from gnnpn_sc_amd.modelPN import CombinatorialRL, reward
    H, T, K = int(fx["hidden"]), int(fx["n_cat"]), int(fx["n_per"])
    nets = []
    for level, seed in (("Low", int(fx["seed_low"])), ("High", int(fx["seed_high"]))):
        m = CombinatorialRL(0, H, T * K, 0, 10, 1, reward, "Dot", K, T, use_cuda=True, level=level)
        m.load_state_dict(opn.make_state_dict(H, seed), strict=True)
        nets.append(m.to(dev).eval())
    return nets


def _inputs(fx):
    return pn_inputs_chunked(int(fx["B"]), int(fx["n_cat"]), int(fx["n_per"]), int(fx["seed_inputs"]), int(fx["chunk"]))


def _slice(fx, lo, hi):
    return {k: fx[k][lo:hi] for k in ("idx_low", "idx_high", "margin_low", "margin_high", "R")}


def _graph_pair(low, high, xs, precision):
    """What bench.py times: the two-level decode of TWO batches as two captured HIP graphs replayed concurrently on two
    streams — decoder build 4 (two workgroups per CU), private workspaces, the LDS footprints PipelinedRunner gives its
    slots (none in f32; one equal footprint of 78 KB in the exact-split precision, pipeline.py)."""
    from gnnpn_sc_amd import ops
    from gnnpn_sc_amd.modelPN import two_level_greedy
    dev = xs[0].device
    streams, graphs, outs, wss = [], [], [], []
    for s, x in enumerate(xs):
        ws = ops.new_workspaces(dev)
        kw = dict(precision=precision, decode_impl=4, lds_kb=78 if precision == "split" else 0, ws=ws)
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            two_level_greedy(low, high, x, **kw)
        torch.cuda.current_stream().wait_stream(st)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = two_level_greedy(low, high, x, **kw)
        ws.frozen = True
        streams.append(st), graphs.append(ops.graph_replay(g, [ws])), outs.append(out), wss.append(ws)
    for _ in range(3):                       # replay several times, both in flight together
        for st, replay in zip(streams, graphs):
            with torch.cuda.stream(st):
                replay()
    for st in streams:
        st.synchronize()
    for ws in wss:
        ws.check()
    return outs


# floors = the measured agreement (tests/golden/agreement_r02.json … agreement_r06.json: 512/512 and 1024/1024 problems identical, no
# flips, on every build and under graph replay): every problem
FLOORS = {"qws512": 1.0, "normal1024": 1.0}


@pytest.mark.parametrize("mode", ["default", "impl4", "graphs-f32", "graphs-split"])
@pytest.mark.parametrize("name,bench_B", [("qws512", 256), ("normal1024", 1024)])
def test_reference_fixture_at_bench_shape(dev, name, bench_B, mode):
    """configs[1] / configs[2] at the bench's batch sizes, against picks the real modelPN.py produced."""
    from gnnpn_sc_amd import ops
    from gnnpn_sc_amd.modelPN import two_level_greedy
    fx = golden(f"pn_big_{name}.npz")
    low, high = _nets(fx, dev)
    x_all = _inputs(fx)
    B = int(fx["B"])
    batches = [(lo, min(B, lo + bench_B)) for lo in range(0, B, bench_B)]
    xs = [x_all[lo:hi].to(dev) for lo, hi in batches]
    if mode.startswith("graphs"):
        pair = xs if len(xs) == 2 else [xs[0], xs[0].clone()]
        outs = _graph_pair(low, high, pair, "split" if mode.endswith("split") else "f32")
        if len(xs) == 1:                                      # the same batch on both slots: both must agree exactly
            for k in ("idx_low", "idx_high", "R"):
                assert torch.equal(outs[0][k], outs[1][k]), k
            outs = outs[:1]
    else:
        outs = [two_level_greedy(low, high, x, decode_impl=4 if mode == "impl4" else 0) for x in xs]
        ops.check_status(dev)
    total = {"problems": 0, "identical_problems": 0, "robust_problems": 0, "robust_identical": 0, "flips": 0,
             "flip_margins": [], "max_R_units": 0}
    for (lo, hi), x, out in zip(batches, xs, outs):
        rec = prefix_parity(out["idx_low"], out["idx_high"], _slice(fx, lo, hi), f"{name}/{mode}[{lo}:{hi}]", x)
        same = rec["same_mask"]
        total["max_R_units"] = max(total["max_R_units"], assert_R_parity(out["R"], fx["R"][lo:hi], f"{name}/{mode}", same))
        assert rec["robust_identical"] == rec["robust_problems"]
        for k in ("problems", "identical_problems", "robust_problems", "robust_identical", "flips"):
            total[k] += rec[k]
        total["flip_margins"] += rec["flip_margins"]
        n_win = fx["win_low"].shape[0]
        if lo < n_win:                                        # window logits of the stored problems
            m = same[: n_win - lo]
            got = out["win_low"].cpu().numpy()[: n_win - lo][m]
            assert np.abs(got - fx["win_low"][lo:n_win][m]).max() < LOGIT_ATOL
            gh = (out["win_high_raw"] + out["win_low"]).cpu().numpy()[: n_win - lo][m]
            assert np.abs(gh - fx["win_high"][lo:n_win][m]).max() < LOGIT_ATOL
    total["agreement"] = total["identical_problems"] / total["problems"]
    record_agreement(f"fixture_{name}_{mode}", total)
    assert total["robust_problems"] >= 150
    assert total["agreement"] >= FLOORS[name], total


MAX_FLIPPED_PROBLEMS = {"synth4": 0, "synth5": 1}       # non-identical problems in the committed agreement record, every build


@pytest.mark.parametrize("mode", ["default", "impl4", "split", "f16-encoder"])
@pytest.mark.parametrize("name", ["synth4", "synth5"])
def test_reference_fixture_long_sequences(dev, name, mode):
    """configs[3] (T=1000, K=5, L=5000) and configs[4] (T=2000, K=10, L=20000): the L-step encoder recurrence, the
    T-step decode and several tiles per cooperative group, against picks of the real modelPN.py.  With 2 x T decisions
    per problem no problem is robust end to end, so the rule is per decision: identical until the first FRAGILE
    decision (prefix_parity).  "f16-encoder" is configs[4]'s opt-in reduced precision: not parity-exact by design, it
    must agree with the reference over at least half of the robust-prefix decisions."""
    from gnnpn_sc_amd import ops
    from gnnpn_sc_amd.modelPN import two_level_greedy
    fx = golden(f"pn_big_{name}.npz")
    low, high = _nets(fx, dev)
    x = _inputs(fx).to(dev)
    kw = {"default": {}, "impl4": {"decode_impl": 4}, "split": {"precision": "split"},
          "f16-encoder": {"precision": "f16"}}[mode]
    out = two_level_greedy(low, high, x, **kw)
    ops.check_status(dev)
    B, T, K = int(fx["B"]), int(fx["n_cat"]), int(fx["n_per"])
    idx = out["idx_high"].cpu().numpy()
    assert ((idx // K) == np.arange(T)[None]).all()                  # every pick inside its step's window
    if mode == "f16-encoder":
        frag = (fx["margin_low"] <= 5e-4) | (fx["margin_high"] <= 5e-4)
        pre = np.where(frag.any(1), frag.argmax(1), T)
        gl, gh = out["idx_low"].cpu().numpy(), idx
        ok = sum(int(((gl[b, :pre[b]] == fx["idx_low"][b, :pre[b]]) & (gh[b, :pre[b]] == fx["idx_high"][b, :pre[b]])).sum())
                 for b in range(B))
        rec = {"robust_prefix_decisions": int(pre.sum()), "identical_in_prefix": ok}
        record_agreement(f"fixture_{name}_{mode}", rec)
        assert ok == int(pre.sum()), rec                                # every robust-prefix decision (measured so since round 2; 64 / 16 problems since round 5)
        return
    rec = prefix_parity(out["idx_low"], out["idx_high"], fx, f"{name}/{mode}", x)
    same = rec["same_mask"]
    rec["max_R_units"] = assert_R_parity(out["R"], fx["R"], f"{name}/{mode}", same)
    n_win = fx["win_low"].shape[0]
    pre = int(np.minimum((fx["margin_low"][:n_win] <= 5e-4).argmax(1), (fx["margin_high"][:n_win] <= 5e-4).argmax(1)).min())
    if pre > 0:                                                        # logits over the common history
        got = out["win_low"].cpu().numpy()[:n_win, :pre]
        assert np.abs(got - fx["win_low"][:, :pre]).max() < LOGIT_ATOL
    record_agreement(f"fixture_{name}_{mode}", rec)
    assert rec["robust_prefix_decisions"] >= 2400                     # 64 / 16 problems since round 5 (16 / 4 before)
    # prefix_parity has already failed on any differing decision with a margin above TAU; what is left to state is how many
    # FRAGILE decisions flipped: none in rounds 2-4 (16 / 4 problems).  With 4 x the problems a flip at a margin of 1e-6 is
    # allowed by the rule and is LISTED (flip_margins in the committed agreement record), not hidden
    assert rec["decisions_compared"] - rec["identical_decisions"] == 2 * (rec["problems"] - rec["identical_problems"]), rec
    # ... and pinned to the committed state (tests/golden/agreement_r05.json): the accounting identity above holds for
    # any number of fragile flips, so a change of summation order that flipped many sub-TAU decisions would pass it — no more
    # non-identical problems than measured (T = 1000: none of 64; T = 2000: ONE of 16, reference margin 1.5e-6), every flip fragile
    assert rec["problems"] - rec["identical_problems"] <= MAX_FLIPPED_PROBLEMS[name], rec
    assert rec["flips"] <= MAX_FLIPPED_PROBLEMS[name] and all(m <= 5e-4 for m in rec["flip_margins"]), rec


def _pipeline(T, S, K, dev, n_gcn, seeds=(7, 8, 9)):
    from gnnpn_sc_amd.modelML import Net
    from gnnpn_sc_amd.modelPN import CombinatorialRL, reward
    from gnnpn_sc_amd.pipeline import ML2PNPipeline
    sd_ml = oml.make_state_dict(128, 20, 2, n_gcn, seed=seeds[0], vocab=max(100, T + 1))
    sd_low, sd_high = opn.make_state_dict(256, seeds[1]), opn.make_state_dict(256, seeds[2])
    net = Net(128, S, 20, 2, n_gcn, vocab=max(100, T + 1))
    net.load_state_dict(sd_ml)
    low = CombinatorialRL(0, 256, T * K, 0, 10, 1, reward, "Dot", K, T, level="Low")
    high = CombinatorialRL(0, 256, T * K, 0, 10, 1, reward, "Dot", K, T, level="High")
    low.load_state_dict(sd_low)
    high.load_state_dict(sd_high)
    return ML2PNPipeline(net.to(dev).eval(), low.to(dev).eval(), high.to(dev).eval(), K), sd_ml, sd_low, sd_high


@pytest.mark.parametrize("T,S,K,B_full,n_ref", [(1000, 5000, 5, 512, 16), (2000, 20000, 10, 256, 8)])   # bench.py's batches; oracle problems 4 -> 16, 2 -> 8 in round 5
def test_whole_pipeline_at_synthetic_sizes(dev, T, S, K, B_full, n_ref):
    """configs[3] / configs[4] through the WHOLE path (vocab > 100 GNN, candidate reduction, both pointer nets) at the
    bench's per-GPU batch, as bench.py runs it (two HIP graphs in flight):
      * the first n_ref problems against the live oracle chain (scores < 1e-5, candidate ids exact, picks by the
        per-decision rule, R in 5-decimal units);
      * size-independent properties over the full batch: picks inside their windows, actions = the picked input rows,
        R recomputed from the actions by the oracle's reward, a problem's result does not depend on its position in
        the batch nor on the batch size (permuted batch and a 16-problem sub-batch give bit-identical rows)."""
    import gnnpn_sc_amd.synth as synth
    from gnnpn_sc_amd.loadData import reduce_from_ranking
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, PipelinedRunner
    pipe, sd_ml, sd_low, sd_high = _pipeline(T, S, K, dev, 2)
    table = synth.make_service_table(T, S, seed=0, degree=32)
    pb = synth.make_problem_batch(table, B_full, seed=3, tasks_per_problem=T)
    svc, batch = DeviceServices.from_table(table, dev), DeviceBatch.from_problems(pb, dev)
    runner = PipelinedRunner(pipe, svc, batch, slots=2)
    outs = []
    for _ in range(2):
        out, s = runner.submit(batch)
        with torch.cuda.stream(runner.stream(s)):
            outs.append({k: out[k].clone() for k in ("idx_low", "idx_high", "R", "actions", "pn_inputs", "candidate_ids", "scores")})
    runner.synchronize(check=True)
    a, b = outs
    for k in a:
        assert torch.equal(a[k], b[k]), k                           # both slots, in flight together: same bits
    # ---- properties over the full batch
    idx = a["idx_high"].long()
    assert bool(((idx // K) == torch.arange(T, device=dev)[None]).all())
    picked = torch.gather(a["pn_inputs"], 1, idx.unsqueeze(-1).expand(-1, -1, 8))
    assert torch.equal(picked, a["actions"])
    R_ref = opn.reward([a["actions"][:, t].cpu() for t in range(T)], "High")
    units = assert_R_parity(a["R"], R_ref, "R recomputed from the actions")
    # ---- position / batch-size independence (bit for bit)
    perm = torch.randperm(B_full, generator=torch.Generator().manual_seed(1))
    pn_ = perm.numpy()
    pbp = synth.ProblemBatch(np.ascontiguousarray(pb.x.reshape(B_full, -1, 7)[pn_].reshape(-1, 7)), pb.edge_index, pb.batch,
                             np.ascontiguousarray(pb.local_bounds[pn_]), np.ascontiguousarray(pb.present[pn_]),
                             np.ascontiguousarray(pb.global_bounds[pn_]))
    outp = eager_reference(pipe, svc, DeviceBatch.from_problems(pbp, dev), decode_impl=runner.decode_impl)
    assert torch.equal(outp["idx_high"], a["idx_high"][perm.to(dev)]) and torch.equal(outp["R"], a["R"][perm.to(dev)])
    sub = synth.ProblemBatch(pb.x[:16 * (T + 1)], pb.edge_index[:, pb.edge_index[0] < 16 * (T + 1)], pb.batch[:16 * (T + 1)],
                             pb.local_bounds[:16], pb.present[:16], pb.global_bounds[:16])
    outs16 = eager_reference(pipe, svc, DeviceBatch.from_problems(sub, dev), decode_impl=runner.decode_impl)
    from gnnpn_sc_amd import ops
    ops.check_status(dev)
    assert torch.equal(outs16["idx_high"], a["idx_high"][:16]) and torch.equal(outs16["R"], a["R"][:16])
    assert torch.equal(outs16["candidate_ids"], a["candidate_ids"][:16])
    # ---- the first n_ref problems against the live oracle chain
    n = n_ref
    nodes = n * (T + 1)
    data = oml.make_data(torch.from_numpy(pb.x[:nodes]), torch.from_numpy(pb.edge_index[:, pb.edge_index[0] < nodes]),
                         torch.from_numpy(pb.batch[:nodes]), torch.from_numpy(table.x_service),
                         torch.from_numpy(table.edge_index), torch.from_numpy(table.edge_attr))
    scores = oml.net_forward(sd_ml, data, 2, 2)
    err = float((a["scores"][:n].cpu() - scores).abs().max())
    assert err < 1e-5, err
    rank = oml.rank_services(scores).numpy()
    # the oracle's own reduction (oracle.data.reduce_candidates, pinned to the real loadDataPN), and — separately — that the
    # product's host helper of the artefact path agrees with it on these problems (the oracle side of this comparison
    # used to be built WITH that helper)
    want_rows = oracle_candidate_rows(pb, table, rank[:n], K, T + 1)
    cat_of = np.repeat(np.arange(T), np.diff(table.cat_ptr))
    helper_rows = torch.tensor([reduce_from_ranking(rank[i], pb.local_bounds[i], pb.present[i], pb.global_bounds[i], cat_of, table.qos, K)
                                for i in range(n)], dtype=torch.float32)[:, :, 1:]
    assert torch.equal(helper_rows, want_rows)
    same_rows = (a["pn_inputs"][:n].cpu() == want_rows).all(-1).all(-1)     # identical candidate reduction (needs the
    assert bool(same_rows.all())                                             # same ranking; measured on all of them)
    keep = same_rows.nonzero().flatten()
    ref = opn.two_level_greedy(sd_low, sd_high, want_rows[keep], T, K)
    rec = prefix_parity(a["idx_low"][:n].cpu()[keep], a["idx_high"][:n].cpu()[keep], {k: ref[k].numpy() for k in
                        ("idx_low", "idx_high", "margin_low", "margin_high")}, f"pipeline T={T}", want_rows[keep])
    rec.update(score_max_err=err, problems_with_identical_candidates=int(same_rows.sum()), of=n, R_recomputed_units=units)
    rec["max_R_units"] = assert_R_parity(a["R"][:n].cpu()[keep], ref["R"], f"pipeline T={T}", rec["same_mask"])
    record_agreement(f"pipeline_T{T}_S{S}_B{B_full}", rec)


def test_normal_full_batch_properties(dev):
    """configs[2] through the whole pipeline at B=1024 (Normal: 4 GCN layers, K=10): the same size-independent
    properties as above plus the first 8 problems against the live oracle chain."""
    import gnnpn_sc_amd.synth as synth
    from gnnpn_sc_amd.loadData import reduce_from_ranking
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, PipelinedRunner
    T, S, K, B, n = 50, 5000, 10, 1024, 8
    pipe, sd_ml, sd_low, sd_high = _pipeline(T, S, K, dev, 4)
    table = synth.make_service_table(T, S, seed=0, degree=32)
    pb = synth.make_problem_batch(table, B, seed=5, tasks_per_problem=10)
    svc, batch = DeviceServices.from_table(table, dev), DeviceBatch.from_problems(pb, dev)
    runner = PipelinedRunner(pipe, svc, batch, slots=2)
    out, s = runner.submit(batch)
    runner.synchronize(check=True)
    idx = out["idx_high"].long()
    assert bool(((idx // K) == torch.arange(T, device=dev)[None]).all())
    assert torch.equal(torch.gather(out["pn_inputs"], 1, idx.unsqueeze(-1).expand(-1, -1, 8)), out["actions"])
    units = assert_R_parity(out["R"], opn.reward([out["actions"][:, t].cpu() for t in range(T)], "High"), "R recomputed")
    nodes = n * 11
    data = oml.make_data(torch.from_numpy(pb.x[:nodes]), torch.from_numpy(pb.edge_index[:, pb.edge_index[0] < nodes]),
                         torch.from_numpy(pb.batch[:nodes]), torch.from_numpy(table.x_service),
                         torch.from_numpy(table.edge_index), torch.from_numpy(table.edge_attr))
    scores = oml.net_forward(sd_ml, data, 2, 4)
    err = float((out["scores"][:n].cpu() - scores).abs().max())
    assert err < 1e-5, err
    rank = oml.rank_services(scores).numpy()
    rows = oracle_candidate_rows(pb, table, rank[:n], K, 11)       # the oracle's reduction, not the product's host helper
    cat_of = np.repeat(np.arange(T), np.diff(table.cat_ptr))
    assert torch.equal(torch.tensor([reduce_from_ranking(rank[i], pb.local_bounds[i], pb.present[i], pb.global_bounds[i], cat_of,
                                                         table.qos, K) for i in range(n)], dtype=torch.float32)[:, :, 1:], rows)
    keep = (out["pn_inputs"][:n].cpu() == rows).all(-1).all(-1).nonzero().flatten()
    assert len(keep) == n                                         # every oracle problem has the oracle's candidate rows
    ref = opn.two_level_greedy(sd_low, sd_high, rows[keep], T, K)
    rec = prefix_parity(out["idx_low"][:n].cpu()[keep], out["idx_high"][:n].cpu()[keep],
                        {k: ref[k].numpy() for k in ("idx_low", "idx_high", "margin_low", "margin_high")}, "normal pipeline", rows[keep])
    rec.update(score_max_err=err, R_recomputed_units=units, problems_with_identical_candidates=int(len(keep)), of=n)
    record_agreement("pipeline_normal_B1024", rec)
