"""-m gpu: every C-ABI entry point against the CPU oracle on the same seeded inputs.
Tolerances: integer/index outputs bit-exact; fp32 outputs within the stated absolute bounds
(different summation order than MKL: ~1e-6 relative per 256-long dot)."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden
from oracle import ml as oml
from oracle import pn as opn

pytestmark = pytest.mark.gpu


def _ops():
    import gnnpn_sc_amd.ops as ops
    import gnnpn_sc_amd.custom_ops  # noqa: F401  (registers torch.ops.gnnpn.* from libgnnpn_torch.so: tests here call both layers)
    return ops


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (37, 50, 26), (256, 128, 24), (300, 1024, 256), (2816, 256, 128),
                                   (4096, 1024, 256), (129, 65, 8), (64, 2507, 128)])
def test_linear_matches_torch(dev, M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = F.linear(a, w, b)
    out = ops.linear(a.to(dev), w.to(dev), b.to(dev)).cpu()
    assert torch.allclose(out, ref, rtol=1e-5, atol=2e-5), float((out - ref).abs().max())
    # epilogue: BN affine + ReLU, and sigmoid
    sc, sh = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    ref2 = torch.relu(ref * sc + sh)
    out2 = ops.linear(a.to(dev), w.to(dev), b.to(dev), sc.to(dev), sh.to(dev), ops.ACT_RELU).cpu()
    assert torch.allclose(out2, ref2, rtol=1e-5, atol=3e-5)
    out3 = ops.linear(a.to(dev), w.to(dev), None, None, None, ops.ACT_SIGMOID).cpu()
    assert torch.allclose(out3, torch.sigmoid(F.linear(a, w)), rtol=1e-5, atol=1e-6)


def test_linear_is_k_ordered_fma_chain(dev):
    """The MFMA accumulation is a k-ordered fp32 fma chain: integer-valued operands are exact."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    a = torch.randint(-8, 9, (70, 256), generator=g).float()
    w = torch.randint(-8, 9, (90, 256), generator=g).float()
    out = ops.linear(a.to(dev), w.to(dev)).cpu()
    assert torch.equal(out, a @ w.t())
    # A = I with asymmetric W catches a transposed C-write
    eye = torch.eye(64)
    w2 = torch.arange(64 * 64, dtype=torch.float32).view(64, 64)
    assert torch.equal(ops.linear(eye.to(dev), w2.to(dev)).cpu(), w2.t())


def test_embed_concat(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    table = torch.randn(100, 20, generator=g)
    x = torch.cat([torch.randint(0, 100, (333, 1), generator=g).float(), torch.rand(333, 6, generator=g)], 1)
    out = ops.embed_concat(x.to(dev), table.to(dev)).cpu()
    assert torch.equal(out, torch.cat([table[x[:, 0].long()], x[:, 1:]], 1))
    bad = x.clone()
    bad[5, 0] = 100
    assert torch.isnan(ops.embed_concat(bad.to(dev), table.to(dev)).cpu()[5, :20]).all()
    wide = torch.randn(100, 40, generator=g)                 # rows wider than 32 columns: the general kernel
    assert torch.equal(ops.embed_concat(x.to(dev), wide.to(dev)).cpu(), torch.cat([wide[x[:, 0].long()], x[:, 1:]], 1))
    big = torch.cat([torch.randint(0, 100, (200003, 1), generator=g).float(), torch.rand(200003, 6, generator=g)], 1)
    assert torch.equal(ops.embed_concat(big.to(dev), table.to(dev)).cpu(), torch.cat([table[big[:, 0].long()], big[:, 1:]], 1))


def _rand_graph(n, e, seed, loops=True):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    if not loops:
        ei = ei[:, ei[0] != ei[1]]
    w = torch.rand(ei.shape[1], generator=g) + 0.05
    return ei, w


@pytest.mark.parametrize("n,e,C", [(5, 8, 2), (50, 400, 26), (301, 3000, 128), (1000, 20000, 256), (200, 0, 256),
                                   (64, 5000, 24)])
def test_gin_aggregate_bit_exact(dev, n, e, C):
    """Sequential CSR-order sums with separate mul/add reproduce scatter_add bit for bit."""
    ops = _ops()
    from gnnpn_sc_amd import graph
    ei, _ = _rand_graph(n, e, n + e)
    x = torch.randn(n, C, generator=torch.Generator().manual_seed(C))
    eps = torch.tensor([0.137])
    ref = oml.scatter_sum(x[ei[0]], ei[1], n) + (1 + eps) * x
    csr = graph.csr_by_destination(ei, n).to(dev)
    out = ops.csr_aggregate(csr.rowptr, csr.col, None, x.to(dev), self_coef=eps.to(dev)).cpu()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("n,e,C", [(50, 400, 24), (301, 3000, 256), (2507, 80000, 256)])
def test_gcn_norm_and_aggregate(dev, n, e, C):
    ops = _ops()
    from gnnpn_sc_amd import graph
    ei, w = _rand_graph(n, e, 11 * n, loops=True)
    x = torch.randn(n, C, generator=torch.Generator().manual_seed(C + 1))
    bias = torch.randn(C, generator=torch.Generator().manual_seed(2))
    row, col, norm = oml.gcn_norm(ei, w, n)
    ref = oml.scatter_sum(norm.view(-1, 1) * x[row], col, n) + bias
    csr = graph.gcn_csr(ei, w, n).to(dev)
    norm_dev = ops.gcn_norm(csr.rowptr, csr.col, csr.w)
    # same multiset of normalised weights, bit for bit (CSR order = stable sort of the oracle's list)
    order = torch.sort(col, stable=True).indices
    assert torch.equal(norm_dev.cpu(), norm[order])
    out = ops.csr_aggregate(csr.rowptr, csr.col, norm_dev, x.to(dev), bias=bias.to(dev)).cpu()
    assert torch.equal(out, ref)


def test_hand_graph_golden(dev):
    """G6: 5-node hand-checkable graph; expected values from the (stand-in) GCNConv run."""
    ops = _ops()
    from gnnpn_sc_amd import graph
    fx = golden("hand_graph.npz")
    x, ei, w = torch.from_numpy(fx["x"]), torch.from_numpy(fx["edge_index"]), torch.from_numpy(fx["w"])
    weight, bias = torch.from_numpy(fx["weight"]), torch.from_numpy(fx["bias"])
    csr = graph.gcn_csr(ei, w, 5).to(dev)
    norm = ops.gcn_norm(csr.rowptr, csr.col, csr.w)
    xw = ops.linear(x.to(dev), weight.t().contiguous().to(dev))
    out = ops.csr_aggregate(csr.rowptr, csr.col, norm, xw, bias=bias.to(dev)).cpu()
    assert torch.allclose(out, torch.from_numpy(fx["gcn"]), rtol=0, atol=1e-6)
    wf = graph.csr_by_destination(ei, 5).to(dev)
    gin = ops.csr_aggregate(wf.rowptr, wf.col, None, x.to(dev), self_coef=torch.tensor([float(fx["gin_eps"])]).to(dev))
    assert torch.equal(gin.cpu(), torch.from_numpy(fx["gin_pre"]))


def test_segment_mean(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    sizes = [3, 1, 11, 0, 7, 48]
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    x = torch.randn(sum(sizes), 128, generator=g)
    ptr = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32)
    out = ops.segment_mean(ptr.to(dev), x.to(dev)).cpu()
    assert torch.equal(out, oml.scatter_mean(x, batch, len(sizes)))


@pytest.mark.parametrize("C", [100, 1, 64, 200])
def test_segment_mean_ragged_channels_and_long_segments(dev, C):
    """One wave per (segment, 64-channel chunk), 8 row loads in flight: channel counts that are no multiple of 64, segments
    longer and shorter than the 8-row unroll, empty segments — the node-order sums of the oracle's scatter_mean, bit for bit."""
    ops = _ops()
    g = torch.Generator().manual_seed(40 + C)
    sizes = [1001, 0, 8, 9, 7, 0, 16, 1, 333]
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    x = torch.randn(sum(sizes), C, generator=g)
    ptr = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32)
    out = ops.segment_mean(ptr.to(dev), x.to(dev)).cpu()
    assert torch.equal(out, oml.scatter_mean(x, batch, len(sizes)))


@pytest.mark.parametrize("S,copies,C,weighted", [(700, 10, 256, True), (2559, 9, 64, True), (333, 40, 128, False)])
def test_csr_aggregate_lds_staged_skewed_degrees(dev, S, copies, C, weighted):
    """The LDS-staged aggregate on block-diagonal copies of a graph with a heavy-tailed degree distribution: rows without
    edges, rows with hundreds of them (many 16-edge batches, the tail groups of 4), the last rows' lists ending at the
    arrays' end (the guarded fetch path), the largest block that takes 16-channel slices (S = 2559) — with the rows dealt by
    descending degree and as they come, against the one-wave-per-row gather kernel: the same bits."""
    from gnnpn_sc_amd import _lib
    from gnnpn_sc_amd._lib import check, dev_ptr, stream_ptr
    ops = _ops()
    g = torch.Generator().manual_seed(S + copies)
    deg = torch.zeros(S, dtype=torch.long)
    heavy = torch.randperm(S, generator=g)[: max(3, S // 40)]
    deg[:] = torch.randint(0, 9, (S,), generator=g)
    deg[heavy] = torch.randint(60, 400, (heavy.numel(),), generator=g)
    deg[torch.randperm(S, generator=g)[: S // 10]] = 0
    deg[S - 1] = 3                                                     # a short list at the very end of the arrays
    rp1 = torch.zeros(S + 1, dtype=torch.long)
    rp1[1:] = torch.cumsum(deg, 0)
    nnz = int(rp1[-1])
    col1 = torch.randint(0, S, (nnz,), generator=g)
    w1 = torch.rand(nnz, generator=g) + 0.25
    rp = torch.cat([rp1[:-1] + c * nnz for c in range(copies)] + [torch.tensor([copies * nnz])]).int().to(dev)
    col = torch.cat([col1 + c * S for c in range(copies)]).int().to(dev)
    w = w1.repeat(copies).to(dev) if weighted else None
    n = copies * S
    x = torch.randn(n, C, generator=g).to(dev)
    bias = torch.randn(C, generator=g).to(dev)
    saved = ops.PREFER_LDS_AGGREGATE
    try:
        ops.PREFER_LDS_AGGREGATE = False
        want = ops.csr_aggregate(rp, col, w, x, bias=bias, act=ops.ACT_RELU)
    finally:
        ops.PREFER_LDS_AGGREGATE = saved
    order = ops.csr_block_row_order(rp, S)
    for o in (order, None):
        y = torch.full_like(x, float("nan"))
        check(_lib.load().gnnpn_csr_aggregate_blocks_f32(
            dev_ptr(rp, torch.int32, "rp"), dev_ptr(col, torch.int32, "col"), dev_ptr(w, torch.float32, "w", True),
            dev_ptr(x, torch.float32, "x"), C, None, dev_ptr(bias, torch.float32, "b"), None, None, ops.ACT_RELU,
            dev_ptr(y, torch.float32, "y"), C, n, C, S, dev_ptr(o, torch.int32, "order", True), stream_ptr()),
            "gnnpn_csr_aggregate_blocks_f32")
        assert torch.equal(y, want)


@pytest.mark.parametrize("H,B,L", [(32, 5, 18), (256, 3, 40), (256, 9, 235)])
def test_lstm_encode_vs_torch(dev, H, B, L):
    ops = _ops()
    sd = opn.make_state_dict(H, 5)
    lstm = opn._lstm_module(sd, "encoder", H)
    x = torch.randn(B, L, H, generator=torch.Generator().manual_seed(9)) * 0.5
    with torch.no_grad():
        ref_out, (ref_h, ref_c) = lstm(x)
    w_ih, b_ih = sd["actor.encoder.weight_ih_l0"], sd["actor.encoder.bias_ih_l0"]
    pre = ops.linear(x.view(B * L, H).to(dev), w_ih.to(dev), b_ih.to(dev)).view(B, L, 4 * H)
    whh = ops.pack_lstm_weight(sd["actor.encoder.weight_hh_l0"]).to(dev)
    net = {"pregates": pre, "whh": whh, "bhh": sd["actor.encoder.bias_hh_l0"].to(dev)}
    enc, h_n, c_n = ops.lstm_encode([net, dict(net)])
    for n in range(2):
        assert float((enc[n].cpu() - ref_out).abs().max()) < 2e-5
        assert float((h_n[n].cpu() - ref_h[0]).abs().max()) < 2e-5
        assert float((c_n[n].cpu() - ref_c[0]).abs().max()) < 5e-5
    assert torch.equal(enc[0], enc[1])      # two nets in one launch are independent and deterministic


@pytest.mark.parametrize("B,L,nets", [(1, 1, 1), (17, 3, 2), (16, 12, 1), (37, 30, 2), (256, 235, 2), (300, 20, 3),
                                      (1040, 9, 2), (530, 7, 4)])
def test_lstm_encode_cooperative_equals_streaming(dev, B, L, nets):
    """The cooperative form (weights in registers, per-step granule hand-off between 8 CUs) and the
    per-workgroup streaming form are the same k-ordered fp32 fma chains: bit-identical outputs.
    Sizes cover partial tiles, several tiles per group (tile switch path) and 3 nets."""
    ops = _ops()
    g = torch.Generator().manual_seed(B + L)
    H = 256
    pre = [(torch.randn(B, L, 4 * H, generator=g) * 0.7).to(dev) for _ in range(nets)]
    whh = [ops.pack_lstm_weight((torch.rand(4 * H, H, generator=g) * 2 - 1) / 16).to(dev) for _ in range(nets)]
    bhh = [((torch.rand(4 * H, generator=g) * 2 - 1) / 16).to(dev) for _ in range(nets)]
    args = [{"pregates": pre[n], "whh": whh[n], "bhh": bhh[n]} for n in range(nets)]
    ref = ops.lstm_encode(args, impl=1)
    out_a = ops.lstm_encode(args, impl=2)   # cooperative form forced
    out = ops.lstm_encode(args)             # default cooperative form: one recurrence per workgroup, 8-CU groups
    out2 = ops.lstm_encode(args)            # back-to-back launches reuse the hand-off buffers
    out3 = ops.lstm_encode(args, write_through=True)   # the placement-independent (agent-scope write-through) hand-off
    ops.check_status(dev)
    for n in range(nets):
        for a, b, c, d, e in zip(ref, out, out2, out3, out_a):
            assert torch.equal(a[n], b[n]) and torch.equal(a[n], c[n]) and torch.equal(a[n], d[n])
            assert torch.equal(a[n], e[n])


def test_in_kernel_input_projection_equals_materialised(dev):
    """inputs . w_in^T + b_in evaluated inside the cooperative encoder is the same k-ordered fma chain
    + bias that gnnpn_linear_f32 materialises: bit-identical encoder outputs."""
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    B, L, H = 50, 33, 256
    x = torch.rand(B, L, 8, generator=g).to(dev)
    w_in = ((torch.rand(4 * H, 8, generator=g) * 2 - 1) * 0.3).to(dev)
    b_in = ((torch.rand(4 * H, generator=g) * 2 - 1) * 0.3).to(dev)
    whh = ops.pack_lstm_weight((torch.rand(4 * H, H, generator=g) * 2 - 1) / 16).to(dev)
    bhh = ((torch.rand(4 * H, generator=g) * 2 - 1) / 16).to(dev)
    pre = ops.linear(x.view(B * L, 8), w_in, b_in).view(B, L, 4 * H)
    a = ops.lstm_encode([{"pregates": pre, "whh": whh, "bhh": bhh}])
    b = ops.lstm_encode([{"inputs": x, "w_in": w_in, "b_in": b_in, "whh": whh, "bhh": bhh}])
    ops.check_status(dev)
    for u, v in zip(a, b):
        assert torch.equal(u[0], v[0])


def test_cell_activations(dev):
    """The hardware-exp2/rcp sigmoid and tanh of the LSTM cells against torch CPU in fp64:
    absolute error at the level of the result's own fp32 rounding, sane at the extremes."""
    ops = _ops()
    x = torch.cat([torch.linspace(-30, 30, 400001), torch.linspace(-0.3, 0.3, 200001),
                   torch.tensor([0.0, -0.0, 1e-30, -1e-30, 88.0, -88.0, 200.0, -200.0, 1e30, -1e30,
                                 float("inf"), float("-inf")])]).float()
    sig, th = (t.cpu().double() for t in ops.debug_cell_activations(x.to(dev)))
    xd = x.double()
    err_s = (sig - torch.sigmoid(xd)).abs().max().item()
    err_t = (th - torch.tanh(xd)).abs().max().item()
    assert err_s < 1.5e-7 and err_t < 1.5e-7, (err_s, err_t)
    small = xd.abs() < 0.25
    rel = ((th - torch.tanh(xd)).abs() / torch.tanh(xd).abs().clamp(min=1e-30))[small & (xd != 0)].max().item()
    assert rel < 3e-7, rel                                   # small arguments keep RELATIVE accuracy
    assert torch.isfinite(sig).all() and torch.isfinite(th).all()
    assert sig[-1] < 1e-37 and sig[-2] == 1.0 and th[-1] == -1.0 and th[-2] == 1.0
    assert th[400001 + 200001] == 0.0 and (sig >= 0).all() and (sig <= 1).all() and (th.abs() <= 1).all()


def test_qos_reward_golden(dev):
    ops = _ops()
    fx = golden("reward.npz")
    a = torch.from_numpy(fx["actions"]).to(dev)
    assert torch.equal(ops.qos_reward(a, "Low").cpu(), torch.from_numpy(fx["R_low"]))
    assert float((ops.qos_reward(a, "High").cpu() - torch.from_numpy(fx["R_high"])).abs().max()) <= 1.0001e-5


@pytest.mark.parametrize("T", [1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 1000])
def test_qos_reward_batches_of_64_rows(dev, T):
    """The reward kernel walks a problem's T action rows in batches of 64 (two alternating row buffers, the chain of a full
    batch unrolled, a shorter last one looped): both sides of every batch boundary against the oracle's reward
    (modelPN.py:15-72), Low (violation counts, exact) and High (rounded to 1e-5)."""
    from oracle import pn as opn
    ops = _ops()
    g = torch.Generator().manual_seed(T)
    B = 9
    act = torch.rand(B, T, 8, generator=g)
    act[:, :, 2:4] = 1.0 - act[:, :, 2:4] * (2.0 / max(T, 2))          # products over T rows that stay near the bounds
    act[:, :, 4:] = 0.0
    act[:, 0, 4:8] = torch.tensor([0.2, 0.9, 0.2, 0.9])
    act[1, :, 0:4] = torch.tensor([0.0, 1.0, 1.0, 1.0])                 # dummy rows except one
    act[1, T // 2, 0] = 0.4
    rows = [act[:, t] for t in range(T)]
    assert torch.equal(ops.qos_reward(act.to(dev), "Low").cpu(), opn.reward(rows, "Low"))
    assert float((ops.qos_reward(act.to(dev), "High").cpu() - opn.reward(rows, "High")).abs().max()) <= 1.0001e-5
    assert len(set(opn.reward(rows, "Low").tolist())) > 1 or T < 3     # the bounds do separate the problems


def test_select_candidates_vs_oracle(dev):
    """Same selections as the oracle's rank-ordered loadDataPN on a dataset in the reference's JSON
    format, including padding and absent categories."""
    import json, os
    from conftest import GOLDEN
    from oracle import data as odata
    from gnnpn_sc_amd import loadData as ld
    ops = _ops()
    fx = json.load(open(os.path.join(GOLDEN, "data_small.json")))
    ds, K, T, S, P = fx["dataset"], fx["K"], fx["T"], fx["S"], fx["P"]
    table, probs = ld.tables_from_dataset(ds)
    rank = torch.tensor(fx["rank_each"])
    # scores that realise the ranking: score = 1 - position/S (exact in fp32 for these sizes)
    scores = torch.empty(P, S)
    scores.scatter_(1, rank, (1.0 - torch.arange(S).float() / S).repeat(P, 1))
    rows, ids = ops.select_candidates(scores.to(dev), torch.from_numpy(table.cat_ptr).to(dev),
                                      torch.from_numpy(table.qos).to(dev),
                                      torch.from_numpy(probs.local_bounds).to(dev),
                                      torch.from_numpy(probs.present).to(dev),
                                      torch.from_numpy(probs.global_bounds).to(dev), K)
    want, _ = odata.load_data_pn(ds["nodefeatures"], ds["serviceFeature"], fx["rank_each"], ds["minCostList"], K)
    want = torch.tensor(want, dtype=torch.float32)[:, :, 1:]
    assert torch.equal(rows.cpu(), want)
    assert (ids.cpu()[(want[:, :, :4] == torch.tensor([0., 1., 1., 1.])).all(-1)] == -1).all()


@pytest.mark.parametrize("K,max_size", [(1, 70), (5, 70), (16, 70), (1, 12), (5, 12), (8, 12), (5, 5), (9, 12)])
def test_select_candidates_16_lane_form_equals_wave_form(dev, K, max_size):
    """n_per <= 16 runs with 16 lanes per (problem, category) segment (four segments per wave) — 8 lanes (eight segments per wave)
    when n_per <= 8 and the categories average at most 8 services, the 1000-task shapes —, larger n_per with a whole
    wave per segment.  The picks are the same sequence, so the first K of a 20-pick selection (emitted cyclically when
    fewer are feasible) are the K-pick selection: categories of 1..max_size services (both sides of the group width, i.e. the
    keys-in-registers path and the strided scan), absent categories, tight and loose bounds, score ties."""
    ops = _ops()
    g = torch.Generator().manual_seed(70 + K + max_size)
    sizes = torch.randint(1, max_size + 1, (37,), generator=g)
    sizes[3], sizes[11] = 1, max_size
    cat_ptr = torch.zeros(38, dtype=torch.int32)
    cat_ptr[1:] = torch.cumsum(sizes, 0).int()
    S, T, B = int(cat_ptr[-1]), 37, 9
    assert (S <= 8 * T) == (max_size <= 12)                              # which launch the case exercises (csrc/select.hip)
    qos = torch.rand(S, 4, generator=g, dtype=torch.float64)
    scores = torch.rand(B, S, generator=g)
    n7 = scores[:, 1::7].shape[1]
    scores[:, 0:7 * n7:7] = scores[:, 1::7]                              # ties -> lowest id
    lo = torch.rand(B, T, 2, generator=g, dtype=torch.float64) * 0.5
    hi = lo + torch.rand(B, T, 2, generator=g, dtype=torch.float64) * 0.7
    local = torch.stack([lo[..., 0], hi[..., 0], lo[..., 1], hi[..., 1]], -1).contiguous()
    local[0, :5] = torch.tensor([0., 1., 0., 1.], dtype=torch.float64)   # everything feasible
    local[1, :5] = torch.tensor([2., 3., 2., 3.], dtype=torch.float64)   # nothing feasible -> dummy rows
    present = (torch.rand(B, T, generator=g) > 0.15).to(torch.uint8)
    glob = torch.rand(B, 4, generator=g, dtype=torch.float64)
    args = (scores.to(dev), cat_ptr.to(dev), qos.to(dev), local.to(dev), present.to(dev), glob.to(dev))
    rows_k, ids_k = ops.select_candidates(*args, K)
    rows_w, ids_w = ops.select_candidates(*args, 20)
    assert torch.equal(ids_k.view(B, T, K), ids_w.view(B, T, 20)[:, :, :K])
    assert torch.equal(rows_k.view(B, T, K, 8)[..., :4], rows_w.view(B, T, 20, 8)[:, :, :K, :4])
    assert torch.equal(rows_k.view(B, T, K, 8)[:, 0, :, 4:], rows_w.view(B, T, 20, 8)[:, 0, :K, 4:])


@pytest.mark.parametrize("B,S", [(3, 40), (4, 2507), (2, 16384), (2, 16385), (3, 20000), (1, 32768)])
def test_rank_rows(dev, B, S):
    ops = _ops()
    g = torch.Generator().manual_seed(S)
    scores = torch.rand(B, S, generator=g)
    scores[:, : S // 4] = scores[:, S // 2: S // 2 + S // 4]      # exact ties -> lowest id first
    out = ops.rank_rows(scores.to(dev)).cpu().long()
    assert torch.equal(out, oml.rank_services(scores))


@pytest.mark.parametrize("B,n_t,n_gin", [(1, 1, 2), (37, 10, 2), (256, 10, 2), (64, 15, 3), (50, 0, 1)])
def test_request_branch_equals_layered(dev, B, n_t, n_gin):
    """gnnpn_request_branch_f32 (the whole GIN branch of small workflow graphs in one launch, node features resident in
    LDS) is stage for stage the arithmetic of embed_concat / csr_aggregate / linear / segment_mean: bit-identical result.
    Ragged sizes: one graph, graphs of 1, 2, 11 and 16 nodes, 1-3 layers."""
    import gnnpn_sc_amd.synth as synth
    from gnnpn_sc_amd.modelML import Net
    from gnnpn_sc_amd.pipeline import DeviceBatch
    from oracle import ml as oml
    T, S = 20, 80
    table = synth.make_service_table(T, S, seed=1, degree=4)
    pb = synth.make_problem_batch(table, B, seed=B + n_t, tasks_per_problem=n_t)
    net = Net(128, S, 20, n_gin, 2)
    net.load_state_dict(oml.make_state_dict(128, 20, n_gin, 2, seed=3))
    net = net.to(dev).eval()
    batch = DeviceBatch.from_problems(pb, dev)
    assert batch.max_nodes == n_t + 1
    fused = net.request_embedding(batch.x, batch.wf_csr, batch.seg_ptr, batch.max_nodes)
    layered = net.request_embedding(batch.x, batch.wf_csr, batch.seg_ptr, 0)
    assert fused.shape == (B, 128) and torch.equal(fused, layered)
    ref = oml.request_embedding(oml.make_state_dict(128, 20, n_gin, 2, seed=3), torch.from_numpy(pb.x),
                                torch.from_numpy(pb.edge_index), torch.from_numpy(pb.batch), B, n_gin)
    assert float((fused.cpu() - ref).abs().max()) < 1e-5
    ops = _ops()
    with pytest.raises(RuntimeError, match="request_branch"):          # graphs beyond 16 nodes are refused, not truncated (C++ operator: c10::Error)
        p = net.prepared(dev)
        flat = [lp[k] for lp in p["gin"] for k in ("w0p", "b0", "a1", "s1", "w3p", "b3", "a2", "s2", "eps")]
        torch.ops.gnnpn.request_branch(batch.x, p["node_table"], batch.wf_csr.rowptr, batch.wf_csr.col, batch.seg_ptr, 17,
                                       flat, p["nodeLin_p"], p["nodeLin"][1], 128)


@pytest.mark.parametrize("S,copies,C,weighted,self_loop", [(40, 3, 256, True, False), (2507, 9, 256, True, False),
                                                            (300, 17, 128, False, True), (5000, 2, 256, True, False),
                                                            (9000, 1, 64, True, False), (97, 5, 24, True, True)])
def test_csr_aggregate_lds_staged_equals_gather(dev, S, copies, C, weighted, self_loop):
    """gnnpn_csr_aggregate_blocks_f32 (node features of a block staged in LDS, one workgroup per (block, channel slice):
    16-, 8- and 4-channel slices by block size) against gnnpn_csr_aggregate_f32 on block-diagonal copies of a random
    weighted graph: bit-identical, with and without edge weights / the GIN self term, ragged last block included."""
    import ctypes
    import gnnpn_sc_amd.synth as synth
    from gnnpn_sc_amd import _lib, graph
    from gnnpn_sc_amd._lib import check, dev_ptr, stream_ptr
    ops = _ops()
    table = synth.make_service_table(5, S, 1, degree=12)
    csr = graph.gcn_csr(torch.from_numpy(table.edge_index), torch.from_numpy(table.edge_attr), S)
    nnz = csr.col.numel()
    n = copies * S - (S // 3 if copies > 1 else 0)                # ragged: the last block is shorter
    rp_full = torch.cat([csr.rowptr[:-1].long() + c * nnz for c in range(copies)] + [torch.tensor([copies * nnz])])
    col_full = torch.cat([csr.col.long() + c * S for c in range(copies)])
    w_full = csr.w.repeat(copies)
    keep = (col_full < n)[: int(rp_full[n])]                      # drop edges whose source fell off the ragged end
    seg = torch.repeat_interleave(torch.arange(n), (rp_full[1:n + 1] - rp_full[:n]))
    cnt = torch.bincount(seg[keep], minlength=n)
    rp = torch.zeros(n + 1, dtype=torch.int32)
    rp[1:] = torch.cumsum(cnt, 0)
    col = col_full[: int(rp_full[n])][keep].int().to(dev)
    w = w_full[: int(rp_full[n])][keep].to(dev) if weighted else None
    rp = rp.to(dev)
    g = torch.Generator().manual_seed(S)
    x = torch.randn(n, C, generator=g).to(dev)
    bias, scale, shift = (torch.randn(C, generator=g).to(dev) for _ in range(3))
    eps = torch.tensor([0.125], device=dev) if self_loop else None
    want = ops.csr_aggregate(rp, col, w, x, self_coef=eps, bias=bias, scale=scale, shift=shift, act=ops.ACT_RELU)
    y = torch.empty_like(x)
    rc = _lib.load().gnnpn_csr_aggregate_blocks_f32(
        dev_ptr(rp, torch.int32, "rp"), dev_ptr(col, torch.int32, "col"), dev_ptr(w, torch.float32, "w", True),
        dev_ptr(x, torch.float32, "x"), C, dev_ptr(eps, torch.float32, "eps", True), dev_ptr(bias, torch.float32, "b"),
        dev_ptr(scale, torch.float32, "s"), dev_ptr(shift, torch.float32, "t"), ops.ACT_RELU, dev_ptr(y, torch.float32, "y"),
        C, n, C, S, None, stream_ptr())
    check(rc, "gnnpn_csr_aggregate_blocks_f32")
    assert torch.equal(y, want)
    # the same with the block's rows dealt by descending degree (gnnpn_csr_block_row_order): a schedule, not a result
    order = ops.csr_block_row_order(rp, S)
    deg = (rp[1:] - rp[:-1]).cpu()
    o = order.cpu().long()
    for b in range(0, n, S):
        blk = o[b:min(b + S, n)]
        assert sorted(blk.tolist()) == list(range(b, min(b + S, n)))             # a permutation of the block's rows
        d = deg[blk]
        assert bool((d[1:] <= d[:-1]).all())                                    # by descending edge count
        same = d[1:] == d[:-1]
        assert bool((blk[1:][same] > blk[:-1][same]).all())                     # ties: lower row first
    y2 = torch.full_like(x, float("nan"))
    rc = _lib.load().gnnpn_csr_aggregate_blocks_f32(
        dev_ptr(rp, torch.int32, "rp"), dev_ptr(col, torch.int32, "col"), dev_ptr(w, torch.float32, "w", True),
        dev_ptr(x, torch.float32, "x"), C, dev_ptr(eps, torch.float32, "eps", True), dev_ptr(bias, torch.float32, "b"),
        dev_ptr(scale, torch.float32, "s"), dev_ptr(shift, torch.float32, "t"), ops.ACT_RELU, dev_ptr(y2, torch.float32, "y"),
        C, n, C, S, dev_ptr(order, torch.int32, "order"), stream_ptr())
    check(rc, "gnnpn_csr_aggregate_blocks_f32")
    assert torch.equal(y2, want)
    rc = _lib.load().gnnpn_csr_aggregate_blocks_f32(
        dev_ptr(rp, torch.int32, "rp"), dev_ptr(col, torch.int32, "col"), None, dev_ptr(x, torch.float32, "x"), C, None, None, None,
        None, 0, dev_ptr(y, torch.float32, "y"), C, n, C, 20000, None, stream_ptr())
    assert rc == -2                                               # a block that cannot fit the LDS is refused (GNNPN_E_UNSUP)


def test_csr_aggregate_picks_the_lds_form_for_block_local_graphs(dev):
    """ops.csr_aggregate with the caller's block_rows promise: the tiled kernel where the plan is valid, the whole-block
    LDS-staged kernel where a block takes 16-channel slices and there are enough (block, slice) workgroups, the gather
    kernel otherwise or when forced — the same bits every way."""
    import gnnpn_sc_amd.synth as synth
    from gnnpn_sc_amd import graph
    ops = _ops()
    S, copies, C = 600, 9, 256
    table = synth.make_service_table(5, S, 3, degree=10)
    csr = graph.gcn_csr(torch.from_numpy(table.edge_index), torch.from_numpy(table.edge_attr), S)
    nnz = csr.col.numel()
    rp = torch.cat([csr.rowptr[:-1].long() + c * nnz for c in range(copies)] + [torch.tensor([copies * nnz])]).int().to(dev)
    col = torch.cat([csr.col.long() + c * S for c in range(copies)]).int().to(dev)
    w = ops.gcn_norm(rp, col, csr.w.repeat(copies).to(dev))
    x = torch.randn(copies * S, C, generator=torch.Generator().manual_seed(1)).to(dev)
    b = torch.randn(C, generator=torch.Generator().manual_seed(2)).to(dev)
    saved, saved_t = ops.PREFER_LDS_AGGREGATE, ops.PREFER_TILED_AGGREGATE
    try:
        ops.PREFER_LDS_AGGREGATE = ops.PREFER_TILED_AGGREGATE = False
        want = ops.csr_aggregate(rp, col, w, x, bias=b, act=ops.ACT_RELU, block_rows=S)          # both LDS forms off: gather
        assert (id(rp), id(col), id(w), S) not in ops._tile_plans and id(rp) not in ops._row_orders
        ops.PREFER_LDS_AGGREGATE = ops.PREFER_TILED_AGGREGATE = None  # defaults: 9 blocks x 16 slices = 144 workgroups
        got = ops.csr_aggregate(rp, col, w, x, bias=b, act=ops.ACT_RELU, block_rows=S)
        # blocks of ONE source tile (<= 2559 rows) qualify for the tiled form in any edge order: it runs first
        assert ops.csr_tile_plan(rp, col, w, S).valid and id(rp) not in ops._row_orders
        assert torch.equal(got, want)
        ops.PREFER_TILED_AGGREGATE = False                            # without it: the whole-block LDS form
        got = ops.csr_aggregate(rp, col, w, x, bias=b, act=ops.ACT_RELU, block_rows=S)
        assert id(rp) in ops._row_orders                              # (it computed and cached the row order)
        assert torch.equal(got, want)
        ops.PREFER_LDS_AGGREGATE = True
        assert torch.equal(ops.csr_aggregate(rp, col, w, x, bias=b, act=ops.ACT_RELU, block_rows=S), want)
        assert torch.equal(ops.csr_aggregate(rp, col, w, x, bias=b, act=ops.ACT_RELU), want)      # no promise: gather form
    finally:
        ops.PREFER_LDS_AGGREGATE, ops.PREFER_TILED_AGGREGATE = saved, saved_t


def _scan_graph_copies(S, copies, degree, seed, dev, ragged=False, self_loops=True):
    """Block-diagonal copies of a graph in the reference's emission order (rows sorted by source; self loops appended by
    gcn_csr, or none), optionally with a shorter last block."""
    import gnnpn_sc_amd.synth as synth
    from gnnpn_sc_amd import graph
    table = synth.make_service_table(5, S, seed, degree=degree, graph="scan")
    ei, ea = torch.from_numpy(table.edge_index), torch.from_numpy(table.edge_attr)
    csr = graph.gcn_csr(ei, ea, S) if self_loops else graph.csr_by_destination(ei, S, ea)
    nnz = csr.col.numel()
    n = copies * S - (S // 3 if ragged else 0)
    rp_full = torch.cat([csr.rowptr[:-1].long() + c * nnz for c in range(copies)] + [torch.tensor([copies * nnz])])
    col_full = torch.cat([csr.col.long() + c * S for c in range(copies)])
    w_full = csr.w.repeat(copies)
    keep = (col_full < n)[: int(rp_full[n])]
    seg = torch.repeat_interleave(torch.arange(n), (rp_full[1:n + 1] - rp_full[:n]))
    rp = torch.zeros(n + 1, dtype=torch.int32)
    rp[1:] = torch.cumsum(torch.bincount(seg[keep], minlength=n), 0)
    return rp.to(dev), col_full[: int(rp_full[n])][keep].int().to(dev), w_full[: int(rp_full[n])][keep].to(dev), n


@pytest.mark.parametrize("S,copies,C,degree,weighted,self_coef,ragged,self_loops", [
    (300, 3, 256, 8, True, False, False, True),        # one source tile, one destination tile, 2 passes
    (2507, 2, 64, 32, True, False, False, True),       # the QWS table: one tile of 2507 rows, 10 passes
    (2600, 3, 32, 12, True, False, True, True),        # two source tiles, two destination tiles, ragged last block
    (5000, 2, 48, 32, True, False, False, True),       # the 1000-task shape: 2 x 2 tiles
    (5000, 1, 16, 6, False, True, False, False),       # no weights, GIN self term, rows without a trailing self loop
    (7000, 1, 32, 20, True, False, False, True),       # 3 x 3 tiles
    (20000, 1, 16, 32, True, False, False, True),      # the 2000-task shape: 8 x 8 tiles
])
def test_csr_aggregate_tiled_equals_gather(dev, S, copies, C, degree, weighted, self_coef, ragged, self_loops):
    """gnnpn_csr_aggregate_tiled_f32 (destination tile x source tile, the source tile's 16-channel slice staged in LDS, the
    edge lists consumed from the plan's sliced-ELL stream) against gnnpn_csr_aggregate_f32 on block-diagonal copies of a
    graph in the reference's edge order: the same bits, with the plan's own accounting checked against the CSR."""
    ops = _ops()
    rp, col, w, n = _scan_graph_copies(S, copies, degree, 3, dev, ragged, self_loops)
    w = w if weighted else None
    g = torch.Generator().manual_seed(S)
    x = torch.randn(n, C, generator=g).to(dev)
    bias, scale, shift = (torch.randn(C, generator=g).to(dev) for _ in range(3))
    eps = torch.tensor([0.125], device=dev) if self_coef else None
    want = ops.csr_aggregate(rp, col, w, x, self_coef=eps, bias=bias, scale=scale, shift=shift, act=ops.ACT_RELU)   # gather form
    x2 = torch.randn(n, C, generator=g).to(dev)
    want2 = ops.csr_aggregate(rp, col, w, x2)
    last = (rp[1:] - 1).clamp(min=0).long()
    loop_row = (col[last] == torch.arange(n, device=dev)).logical_and(rp[1:] > rp[:-1])      # rows that end with their self loop
    plan = ops.csr_tile_plan(rp, col, w, S)
    assert plan.valid, plan.stats
    st, gm = plan.stats, plan.geom
    assert gm["src_tiles"] == -(-S // 2559) and gm["dst_tiles"] == -(-S // 2560)
    # the stream holds every edge but the trailing self loops of rows OUTSIDE the last source tile (the epilogue adds those)
    own_tile = (torch.arange(n, device=dev) % S) // gm["src_tile_rows"]
    n_epilogue = int((loop_row & (own_tile != gm["src_tiles"] - 1)).sum())
    assert st["edges"] == col.numel() - n_epilogue and st["rows"] == n
    assert st["slots"] >= st["edges"] and st["quads"] * 64 == st["slots"] and sum(st["run_histogram"]) == n * gm["src_tiles"]
    got = plan.aggregate(x, eps, bias, scale, shift, ops.ACT_RELU)
    assert torch.equal(got, want)
    assert ops.csr_tile_plan(rp, col, w, S) is plan                               # cached per (graph, weights)
    assert torch.equal(plan.aggregate(x2), want2)                                 # a second layer on the same plan, no epilogue
    # the kernel walks a unit's quads four at a time or (blocks of three or more source tiles) two at a time: both walks, every case
    for walk in ("quads", "pairs"):
        os.environ["GNNPN_TILED_WALK"] = walk
        try:
            assert torch.equal(plan.aggregate(x, eps, bias, scale, shift, ops.ACT_RELU), want), walk
            assert torch.equal(plan.aggregate(x2), want2), walk
        finally:
            del os.environ["GNNPN_TILED_WALK"]
    # the stream ends with FOUR quads of slack: an empty unit at the end of the scan order (5000 rows per block: the second
    # destination tile has 2488 rows, its last unit none) starts AT the stream's end and still requests its first four quads —
    # loaded, never used (three quads of slack left that read 512 B out of bounds)
    assert plan.batches.numel() == (st["quads"] + 4) * 512
    plan.batches[st["quads"] * 512:] = 0xFF
    assert torch.equal(plan.aggregate(x2), want2)
    plan.batches[st["quads"] * 512:] = 0


def test_csr_aggregate_tiled_skewed_degrees_and_invalid_graphs(dev):
    """Heavy-tailed degrees on the tiled form — empty rows, hub rows with more than a thousand edges inside one source tile
    (more quads than the sort key holds: only the order of the units depends on it) — and the validity check: a graph whose
    rows are not in source order is refused by the plan (ops.csr_aggregate then keeps the gather form)."""
    ops = _ops()
    S, C = 5200, 32
    g = torch.Generator().manual_seed(11)
    deg = torch.randint(0, 9, (S,), generator=g)
    heavy = torch.randperm(S, generator=g)[:40]
    deg[heavy] = torch.randint(200, 700, (40,), generator=g)
    deg[heavy[:3]] = torch.tensor([2300, 1500, 1100])
    deg[torch.randperm(S, generator=g)[: S // 10]] = 0
    rows = []
    for i in range(S):
        c = torch.randperm(S, generator=g)[: int(deg[i])].sort().values       # sorted sources, no duplicates
        c = c[c != i]
        rows.append(torch.cat([c, torch.tensor([i])]) if i % 7 else c)         # most rows end with their self loop
    rp = torch.zeros(S + 1, dtype=torch.int32)
    rp[1:] = torch.cumsum(torch.tensor([r.numel() for r in rows]), 0)
    col = torch.cat(rows).int().to(dev)
    rp = rp.to(dev)
    w = (torch.rand(col.numel(), generator=g) + 0.25).to(dev)
    x = torch.randn(S, C, generator=g).to(dev)
    want = ops.csr_aggregate(rp, col, w, x)
    plan = ops.csr_tile_plan(rp, col, w, S)
    assert plan.valid and plan.stats["run_histogram"][63] >= 40
    assert torch.equal(plan.aggregate(x), want)
    # the same lists with one row's sources out of order: invalid, and the dispatcher keeps the gather form
    col_bad = col.clone()
    r = int(heavy[5])
    lo, hi = int(rp[r]), int(rp[r + 1])
    col_bad[lo:hi - 1] = col_bad[lo:hi - 1].flip(0)
    bad = ops.csr_tile_plan(rp, col_bad, w, S)
    assert not bad.valid and bad.stats["invalid_rows"] == 1
    saved = ops.PREFER_TILED_AGGREGATE
    try:
        ops.PREFER_TILED_AGGREGATE = True
        assert torch.equal(ops.csr_aggregate(rp, col_bad, w, x, block_rows=S), ops.csr_aggregate(rp, col_bad, w, x))
        assert torch.equal(ops.csr_aggregate(rp, col, w, x, block_rows=S), want)       # valid plan: the tiled form, same bits
    finally:
        ops.PREFER_TILED_AGGREGATE = saved


@pytest.mark.parametrize("S,copies,degree,weighted,ragged,self_loops,shuffle", [
    (300, 3, 8, True, False, True, False),          # one source tile, one destination tile
    (2600, 2, 12, True, True, True, False),         # two source and two destination tiles, a shorter last block
    (5000, 2, 32, True, False, True, False),        # the 1000-task shape: 2 x 2 tiles, self loops in and out of the last tile
    (5200, 1, 6, False, False, False, False),       # three source tiles, no weights, no self loops
    (5000, 1, 16, True, False, True, True),         # edge lists in another order: rows that break the tile order are counted
])
def test_tile_plan_equals_the_numpy_restatement(dev, S, copies, degree, weighted, ragged, self_loops, shuffle):
    """The plan of the tiled aggregate is integer work — tile geometry, the per-row validity rule (does the row's list visit the
    source tiles in order?), the rows' order inside their units, quads per (unit, source tile), the sliced-ELL stream of LDS
    offsets and weight bits: bit for bit the plan of oracle/tile_plan.py, an independent numpy restatement."""
    from oracle import tile_plan as otp
    ops = _ops()
    rp, col, w, n = _scan_graph_copies(S, copies, degree, 11, dev, ragged=ragged, self_loops=self_loops)
    if shuffle:                                                   # reverse every other row's list (self loop kept last)
        rpc, colc = rp.cpu(), col.cpu().clone()
        for r in range(0, n, 2):
            lo, hi = int(rpc[r]), int(rpc[r + 1]) - 1
            colc[lo:hi] = colc[lo:hi].flip(0)
        col = colc.to(dev)
    wd = w if weighted else None
    plan = ops.TilePlan(rp, col, wd, S)
    ref = otp.build(rp.cpu().numpy(), col.cpu().numpy(), None if wd is None else wd.cpu().numpy(), n, S)
    g = ref["geometry"]
    assert plan.geom == {"n_blocks": g["n_blocks"], "src_tiles": g["NT"], "src_tile_rows": g["TR"], "dst_tiles": g["ND"],
                         "dst_tile_rows": g["DR"], "units": g["U"], "wavefronts": otp.WAVES, "passes": g["passes"]}
    for k in ("invalid_rows", "edges", "slots", "rows", "run_histogram"):
        assert plan.stats[k] == ref["stats"][k], k
    assert plan.valid == ref["valid"] == (not shuffle)
    assert np.array_equal(plan.order.cpu().numpy(), ref["order"])
    assert np.array_equal(plan.selfw.view(torch.int32).cpu().numpy().view(np.uint32), ref["selfw_bits"])
    assert np.array_equal(plan.header.cpu().numpy().reshape(-1, 2)[:, 1], ref["header"][:, 1])          # quads per (unit, source tile)
    if plan.valid:
        assert plan.stats["quads"] == ref["stats"]["quads"]
        assert np.array_equal(plan.header.cpu().numpy().reshape(-1, 2), ref["header"])
        assert np.array_equal(plan.batches.cpu().numpy().view(np.uint32), ref["batches"])


def test_csr_aggregate_dispatch_prefers_the_tiled_form_on_the_reference_edge_order(dev):
    """ops.csr_aggregate with the block_rows promise on copies of a graph in the reference's emission order: the tiled form runs
    (its plan is built once and cached), results equal the gather form's; through Net.service_embedding the normalised weights
    — and so the plan — are computed once per CSR object."""
    from gnnpn_sc_amd import graph
    ops = _ops()
    S, copies, C = 2507, 9, 256
    rp, col, w_raw, n = _scan_graph_copies(S, copies, 32, 5, dev)
    w = ops.gcn_norm(rp, col, w_raw)
    x = torch.randn(n, C, generator=torch.Generator().manual_seed(1)).to(dev)
    b = torch.randn(C, generator=torch.Generator().manual_seed(2)).to(dev)
    want = ops.csr_aggregate(rp, col, w, x, bias=b, act=ops.ACT_RELU)                        # no promise: gather form
    before = len(ops._tile_plans)
    got = ops.csr_aggregate(rp, col, w, x, bias=b, act=ops.ACT_RELU, block_rows=S)           # 9 x 1 x 16 = 144 workgroups: tiled
    assert len(ops._tile_plans) == before + 1 and ops.csr_tile_plan(rp, col, w, S).valid
    assert torch.equal(got, want)
    saved = ops.PREFER_TILED_AGGREGATE
    try:
        ops.PREFER_TILED_AGGREGATE = False
        assert torch.equal(ops.csr_aggregate(rp, col, w, x, bias=b, act=ops.ACT_RELU, block_rows=S), want)   # whole-block LDS form
    finally:
        ops.PREFER_TILED_AGGREGATE = saved


@pytest.mark.parametrize("n,c_in,nodes_per_graph,with_lin", [(5000, 26, 1001, False), (5000, 128, 1001, True), (70, 26, 7, True),
                                                             (4097, 128, 50, False), (300, 24, 11, True)])
def test_gin_layer_equals_the_separate_kernels(dev, n, c_in, nodes_per_graph, with_lin):
    """gnnpn_gin_layer_f32 (aggregate -> Linear + BN + ReLU -> Linear + BN + ReLU [-> nodeLin] in one launch, the [rows x 256]
    intermediate in LDS) against gnnpn_csr_aggregate_f32 + gnnpn_linear_f32 x 2 (+ 1): the same fma chains, zero padding and
    epilogue roundings -> the same bits; chain graphs with a ragged last row tile, a last graph cut short, isolated nodes."""
    from gnnpn_sc_amd import graph
    ops = _ops()
    g = torch.Generator().manual_seed(n + c_in)
    i = torch.arange(n - 1)
    keep = (i + 1) % nodes_per_graph != 0                                  # chain edges inside each graph, both directions
    src = torch.stack([i[keep], i[keep] + 1], 1).reshape(-1)
    dst = torch.stack([i[keep] + 1, i[keep]], 1).reshape(-1)
    extra = torch.randint(0, n, (2, n // 3), generator=g)                   # and some long-range edges (rows far outside the tile)
    ei = torch.cat([torch.stack([src, dst]), extra], 1)
    csr = graph.csr_by_destination(ei, n).to(dev)
    x = torch.randn(n, c_in, generator=g).to(dev)
    eps = torch.tensor([0.07], device=dev)
    mk = lambda *s: (torch.randn(*s, generator=g) / s[-1] ** 0.5).to(dev)   # noqa: E731
    w1, b1, w2, b2, w3, b3 = mk(256, c_in), mk(256), mk(128, 256), mk(128), mk(128, 128), mk(128)
    a1, s1, a2, s2 = (torch.rand(256, generator=g) + 0.5).to(dev), mk(256), (torch.rand(128, generator=g) + 0.5).to(dev), mk(128)
    agg = ops.csr_aggregate(csr.rowptr, csr.col, None, x, self_coef=eps)
    t = ops.linear(agg, w1, b1, a1, s1, ops.ACT_RELU)
    want = ops.linear(t, w2, b2, a2, s2, ops.ACT_RELU)
    if with_lin:
        want = ops.linear(want, w3, b3)
    q1, q2, q3 = ops.pack_mfma_b32(w1), ops.pack_mfma_b32(w2), ops.pack_mfma_b32(w3)          # the matrix core's B-fragment layout
    got = ops.gin_layer(csr.rowptr, csr.col, x, eps, q1, b1, a1, s1, q2, b2, a2, s2, q3 if with_lin else None, b3 if with_lin else None)
    assert got.shape == want.shape and torch.equal(got, want)
    got = ops.gin_layer(csr.rowptr, csr.col, x, eps, q1, None, None, None, q2, b2, None, None)      # no bias / no BN operands
    assert torch.equal(got, ops.linear(ops.linear(agg, w1, None, None, None, ops.ACT_RELU), w2, b2, None, None, ops.ACT_RELU))
    with pytest.raises(ops.GnnpnError):
        ops.gin_layer(csr.rowptr, csr.col, x, eps, ops.pack_mfma_b32(w1[:128]), b1[:128], a1[:128], s1[:128],
                      ops.pack_mfma_b32(mk(128, 128)), b2, a2, s2)                                   # h1 != 256
    with pytest.raises(ops.GnnpnError):
        ops.gin_layer(csr.rowptr, csr.col, x, eps, w1, b1, a1, s1, w2, b2, a2, s2)                  # unpacked weights


def _unpack_split_weights(packed, inv, N, K):
    """The fp32 weight an image of ops.pack_split_weights stands for: (p0 + p1 / 2^11 + p2 / 2^22) * col_inv, in float64."""
    kb = (K + 31) // 32
    rec = packed.cpu().numpy().reshape(N // 16, kb, 2560)
    p0 = rec[:, :, :1024].copy().view(np.float16).reshape(N // 16, kb, 64, 8).astype(np.float64)
    p1 = rec[:, :, 1024:2048].copy().view(np.float16).reshape(N // 16, kb, 64, 8).astype(np.float64)
    b2 = rec[:, :, 2048:].copy().reshape(N // 16, kb, 64, 8)
    p2 = (b2.astype(np.uint16) << 8).view(np.float16).astype(np.float64)
    v = (p0 + p1 / 2048.0 + p2 / 2048.0 ** 2).reshape(N // 16, kb, 4, 16, 8)        # [t, kk, kq, c, j]
    w = v.transpose(0, 3, 1, 2, 4).reshape(N, kb * 32)[:, :K]
    return w * inv.double().cpu().numpy()[:, None]


@pytest.mark.parametrize("n,c_in,nodes_per_graph,with_lin", [(5000, 26, 1001, False), (5000, 128, 1001, True), (70, 26, 7, True),
                                                             (4097, 128, 50, False), (300, 24, 11, True), (333, 64, 40, True)])
def test_gin_layer_split_against_fp64(dev, n, c_in, nodes_per_graph, with_lin):
    """gnnpn_gin_layer_split — the layer's three dense products on the fp16 matrix cores through the exact 3-piece split —
    against an fp64 evaluation of the layer (from the SAME fp32 aggregate; intermediates rounded to fp32 where both kernels
    round them), next to the fp32 layer's error: the split build must be at least as close (factor 1.25 of slack on the
    maximum, none needed so far), and the packed weight pieces must sum to the weights bit for bit."""
    from gnnpn_sc_amd import graph
    ops = _ops()
    g = torch.Generator().manual_seed(n + c_in)
    i = torch.arange(n - 1)
    keep = (i + 1) % nodes_per_graph != 0
    src = torch.stack([i[keep], i[keep] + 1], 1).reshape(-1)
    dst = torch.stack([i[keep] + 1, i[keep]], 1).reshape(-1)
    extra = torch.randint(0, n, (2, n // 3), generator=g)
    csr = graph.csr_by_destination(torch.cat([torch.stack([src, dst]), extra], 1), n).to(dev)
    x = (torch.randn(n, c_in, generator=g) * torch.exp(2 * torch.randn(n, 1, generator=g))).to(dev)   # rows of very different magnitude
    x[5] = 0.0
    eps = torch.tensor([0.07], device=dev)
    mk = lambda *s: (torch.randn(*s, generator=g) / s[-1] ** 0.5).to(dev)   # noqa: E731
    w1, b1, w2, b2, w3, b3 = mk(256, c_in), mk(256), mk(128, 256), mk(128), mk(128, 128), mk(128)
    a1, s1, a2, s2 = (torch.rand(256, generator=g) + 0.5).to(dev), mk(256), (torch.rand(128, generator=g) + 0.5).to(dev), mk(128)
    p1, p2, p3 = ops.pack_split_weights(w1), ops.pack_split_weights(w2), ops.pack_split_weights(w3)
    for (pk, inv), w in ((p1, w1), (p2, w2), (p3, w3)):
        assert np.array_equal(_unpack_split_weights(pk, inv, *w.shape), w.double().cpu().numpy())
    got = ops.gin_layer_split(csr.rowptr, csr.col, x, eps, *p1, b1, a1, s1, *p2, b2, a2, s2,
                              *(p3 if with_lin else (None, None)), b3 if with_lin else None)
    f32 = ops.gin_layer(csr.rowptr, csr.col, x, eps, ops.pack_mfma_b32(w1), b1, a1, s1, ops.pack_mfma_b32(w2), b2, a2, s2,
                        ops.pack_mfma_b32(w3) if with_lin else None, b3 if with_lin else None)
    agg = ops.csr_aggregate(csr.rowptr, csr.col, None, x, self_coef=eps).double()
    D = lambda t: t.double()   # noqa: E731
    t = torch.relu((agg @ D(w1).t() + D(b1)) * D(a1) + D(s1)).float().double()           # both kernels hold T in fp32
    ref = torch.relu((t @ D(w2).t() + D(b2)) * D(a2) + D(s2))
    if with_lin:
        ref = ref.float().double() @ D(w3).t() + D(b3)
    assert got.shape == f32.shape == ref.shape
    scale = ref.abs().amax(1, keepdim=True).clamp_min(1e-30)                               # per row: rows differ by orders of magnitude
    e_split, e_f32 = ((got.double() - ref).abs() / scale), ((f32.double() - ref).abs() / scale)
    assert torch.isfinite(got).all()
    assert e_split.max().item() <= 1.25 * e_f32.max().item() + 1e-7, (e_split.max().item(), e_f32.max().item())
    assert e_split.mean().item() <= 1.25 * e_f32.mean().item() + 1e-9, (e_split.mean().item(), e_f32.mean().item())
    assert e_split.max().item() < 5e-6
    with pytest.raises(ops.GnnpnError):
        ops.gin_layer_split(csr.rowptr, csr.col, x, eps, *p2, b1, a1, s1, *p2, b2, a2, s2)     # weights that do not chain
    assert not ops.gin_layer_split_supported(256, 256, 128) and ops.gin_layer_split_supported(c_in, 256, 128, 128)


@pytest.mark.parametrize("n", [0, 1, 47, 48, 49, 97])
def test_gin_layer_split_row_tile_boundaries(dev, n):
    """Empty input, a single row, and row counts on both sides of the 48-row workgroup tile: the split layer against the fp32
    layer (same aggregate, same epilogues; the products differ by their summation order only)."""
    from gnnpn_sc_amd import graph
    ops = _ops()
    g = torch.Generator().manual_seed(100 + n)
    c_in = 128
    mk = lambda *s: (torch.randn(*s, generator=g) / s[-1] ** 0.5).to(dev)   # noqa: E731
    w1, b1, w2, b2, w3, b3 = mk(256, c_in), mk(256), mk(128, 256), mk(128), mk(128, 128), mk(128)
    a1, s1, a2, s2 = (torch.rand(256, generator=g) + 0.5).to(dev), mk(256), (torch.rand(128, generator=g) + 0.5).to(dev), mk(128)
    ei = torch.stack([torch.arange(max(n - 1, 0)), torch.arange(1, max(n, 1))]) if n > 1 else torch.zeros((2, 0), dtype=torch.long)
    csr = graph.csr_by_destination(ei, n).to(dev)
    x = torch.randn(n, c_in, generator=g).to(dev)
    eps = torch.tensor([0.3], device=dev)
    p1, p2, p3 = ops.pack_split_weights(w1), ops.pack_split_weights(w2), ops.pack_split_weights(w3)
    got = ops.gin_layer_split(csr.rowptr, csr.col, x, eps, *p1, b1, a1, s1, *p2, b2, a2, s2, *p3, b3)
    want = ops.gin_layer(csr.rowptr, csr.col, x, eps, ops.pack_mfma_b32(w1), b1, a1, s1, ops.pack_mfma_b32(w2), b2, a2, s2,
                         ops.pack_mfma_b32(w3), b3)
    assert got.shape == want.shape == (n, 128)
    if n:
        assert torch.isfinite(got).all() and float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    via_op = torch.ops.gnnpn.gin_layer_split(csr.rowptr, csr.col, x, eps, *p1, b1, a1, s1, *p2, b2, a2, s2, *p3, b3)   # the C++ operator
    assert torch.equal(via_op, got)


def test_gcn_layer_against_dense_fp64_formula(dev):
    """The HIP GCN layer (gcn_csr + gcn_norm + linear + csr_aggregate) against the dense float64 matrix formula
    D^-1/2 (A_w + I) D^-1/2 X W + b — an oracle-independent check of the arithmetic whose reference implementation
    (torch_geometric 1.7.0) is not available to pin against."""
    from gnnpn_sc_amd import graph
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    n, e, cin, cout = 300, 4000, 24, 256
    ei = torch.randint(0, n, (2, e), generator=g)
    ei = ei[:, ei[0] != ei[1]]
    w = torch.rand(ei.shape[1], generator=g) + 0.1
    x = torch.randn(n, cin, generator=g)
    W = torch.randn(cin, cout, generator=g) * 0.3
    b = torch.randn(cout, generator=g)
    A = torch.zeros(n, n, dtype=torch.float64)
    A.index_put_((ei[1], ei[0]), w.double(), accumulate=True)
    A += torch.eye(n, dtype=torch.float64)
    dis = A.sum(1).pow(-0.5)
    want = (dis[:, None] * A * dis[None, :]) @ (x.double() @ W.double()) + b.double()
    csr = graph.gcn_csr(ei, w, n).to(dev)
    norm = ops.gcn_norm(csr.rowptr, csr.col, csr.w)
    xw = ops.linear(x.to(dev), W.t().contiguous().to(dev))
    got = ops.csr_aggregate(csr.rowptr, csr.col, norm, xw, bias=b.to(dev))
    assert float((got.double().cpu() - want).abs().max()) < 5e-5


def test_empty_inputs_everywhere(dev):
    """Zero rows / zero problems / zero edges through every op of the path: empty outputs of the right shape, no launch
    error, nothing written out of bounds."""
    ops = _ops()
    from gnnpn_sc_amd import graph
    z = lambda *s: torch.zeros(s, device=dev)   # noqa: E731
    assert ops.linear(z(0, 8), torch.rand(5, 8, device=dev), torch.rand(5, device=dev)).shape == (0, 5)
    assert ops.embed_concat(z(0, 7), torch.rand(100, 20, device=dev)).shape == (0, 26)
    assert ops.rank_rows(z(0, 9)).shape == (0, 9)
    assert ops.qos_reward(z(0, 4, 8), "High").shape == (0,)
    # a graph without edges: the aggregate is the self term only
    x = torch.rand(6, 12, device=dev)
    csr = graph.csr_by_destination(torch.zeros((2, 0), dtype=torch.long, device=dev), 6)
    eps = torch.tensor([0.25], device=dev)
    y = ops.csr_aggregate(csr.rowptr, csr.col, None, x, self_coef=eps)
    assert torch.allclose(y, 1.25 * x, rtol=0, atol=1e-6)
    assert ops.segment_mean(torch.tensor([0, 0, 6], dtype=torch.int32, device=dev), x).shape == (2, 12)   # an empty segment -> zeros
    assert float(ops.segment_mean(torch.tensor([0, 0, 6], dtype=torch.int32, device=dev), x)[0].abs().max()) == 0.0
    H = 256
    net = {"inputs": z(0, 7, 8), "w_in": torch.rand(4 * H, 8, device=dev), "b_in": torch.rand(4 * H, device=dev),
           "whh": ops.pack_lstm_weight(torch.rand(4 * H, H) / 16).to(dev), "bhh": torch.rand(4 * H, device=dev)}
    enc, h_n, c_n = ops.lstm_encode([net])
    assert enc[0].shape == (0, 7, H) and h_n[0].shape == (0, H)
    ops.check_status(dev)
