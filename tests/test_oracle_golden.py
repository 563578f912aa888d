"""-m "not gpu": the oracle (oracle/) against the golden vectors generated from the imported
reference (tests/golden/make_golden.py).  This is what pins the checker."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden
from oracle import data as odata
from oracle import ml as oml
from oracle import pn as opn


@pytest.mark.parametrize("name", ["small", "dummy", "qws", "normal", "saturated"])
def test_pn_oracle_reproduces_reference(name):
    fx = golden(f"pn_{name}.npz")
    torch.set_num_threads(1)
    H, T, K = int(fx["hidden"]), int(fx["n_cat"]), int(fx["n_per"])
    sds = [opn.make_state_dict(H, int(fx["seed_low"])), opn.make_state_dict(H, int(fx["seed_high"]))]
    scale = float(fx["weight_scale"]) if "weight_scale" in fx.files else 1.0
    for sd in sds:
        for k in sd:
            if scale != 1.0 and ("encoder" in k or "decoder." in k):
                sd[k] = sd[k] * scale
    out = opn.two_level_greedy(sds[0], sds[1], torch.from_numpy(fx["inputs"]), T, K)
    assert np.array_equal(out["idx_low"].numpy(), fx["idx_low"])
    assert np.array_equal(out["idx_high"].numpy(), fx["idx_high"])
    assert np.array_equal(out["actions"].numpy(), fx["actions"])
    assert np.allclose(out["R"].numpy(), fx["R"], rtol=0, atol=1e-6)
    assert np.allclose(out["win_low"].numpy(), fx["win_low"], rtol=0, atol=2e-5)
    assert np.allclose(out["win_high"].numpy(), fx["win_high"], rtol=0, atol=2e-5)
    lat1 = out["latent"][1].numpy()
    assert np.array_equal(np.isinf(lat1), np.isinf(fx["latent_step1"]))


ATTN_FIXTURES = ["dot_g1_small", "bahdanau_g0_small", "bahdanau_g2_small", "dot_g1_qws", "bahdanau_g1_qws"]


@pytest.mark.parametrize("name", ATTN_FIXTURES)
def test_pn_oracle_attention_forms_reproduce_reference(name):
    """pn_attn_*.npz: the real modelPN.py with attention='Bahdanau' and / or n_glimpses > 0."""
    fx = golden(f"pn_attn_{name}.npz")
    torch.set_num_threads(1)
    H, T, K = int(fx["hidden"]), int(fx["n_cat"]), int(fx["n_per"])
    att, ng = str(fx["attention"]), int(fx["n_glimpses"])
    sds = [opn.make_state_dict(H, int(fx["seed_low"]), attention=att), opn.make_state_dict(H, int(fx["seed_high"]), attention=att)]
    out = opn.two_level_greedy(sds[0], sds[1], torch.from_numpy(fx["inputs"]), T, K, attention=att, n_glimpses=ng)
    for key in ("idx_low", "idx_high", "actions"):
        assert np.array_equal(out[key].numpy(), fx[key]), key
    assert np.allclose(out["R"].numpy(), fx["R"], rtol=0, atol=1e-6)
    assert np.allclose(out["win_low"].numpy(), fx["win_low"], rtol=0, atol=2e-5)
    assert np.allclose(out["win_high"].numpy(), fx["win_high"], rtol=0, atol=2e-5)
    # the Bahdanau parameters are drawn after everything else: the 'Dot' weights of the same seed are unchanged
    plain = opn.make_state_dict(H, int(fx["seed_low"]))
    assert all(torch.equal(plain[k], sds[0][k]) for k in plain)


@pytest.mark.parametrize("name", ["small", "qws"])
def test_pn_oracle_category_embedding_reproduces_reference(name):
    """pn_embed_*.npz: the real modelPN.py with embedding_size != 0 (embeddingTag=1; modelPN.py:153-154,183-188, reward's
    tag :42-45): rows [category | 8 floats]."""
    fx = golden(f"pn_embed_{name}.npz")
    torch.set_num_threads(1)
    H, T, K, E = int(fx["hidden"]), int(fx["n_cat"]), int(fx["n_per"]), int(fx["embedding_size"])
    sds = [opn.make_state_dict(H, int(fx[k]), embedding_size=E, n_cat=T) for k in ("seed_low", "seed_high")]
    assert fx["inputs"].shape[2] == 9 and sds[0]["actor.embedding2.weight"].shape == (H, E + 8)
    out = opn.two_level_greedy(sds[0], sds[1], torch.from_numpy(fx["inputs"]), T, K)
    for key in ("idx_low", "idx_high", "actions"):
        assert np.array_equal(out[key].numpy(), fx[key]), key
    assert out["actions"].shape[2] == 9                                   # the action rows keep the category column (:293-295)
    assert np.allclose(out["R"].numpy(), fx["R"], rtol=0, atol=1e-6)
    assert np.allclose(out["win_low"].numpy(), fx["win_low"], rtol=0, atol=2e-5)
    assert np.allclose(out["win_high"].numpy(), fx["win_high"], rtol=0, atol=2e-5)


def test_lstm_cell_explicit_matches_nn_lstm():
    """Documents the cell arithmetic the kernels implement (gate order i,f,g,o)."""
    sd = opn.make_state_dict(32, 3)
    lstm = opn._lstm_module(sd, "decoder", 32)
    g = torch.Generator().manual_seed(0)
    x, h, c = (torch.randn(5, 32, generator=g) for _ in range(3))
    with torch.no_grad():
        _, (h_ref, c_ref) = lstm(x.unsqueeze(1), (h.unsqueeze(0), c.unsqueeze(0)))
    h2, c2 = opn.lstm_cell_explicit(x, h, c, *(sd[f"actor.decoder.{k}"] for k in
                                               ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")))
    assert torch.allclose(h2, h_ref[0], atol=1e-6) and torch.allclose(c2, c_ref[0], atol=1e-6)


def test_reward_known_answers():
    fx = golden("reward.npz")
    actions = [torch.from_numpy(fx["actions"][:, t]) for t in range(fx["actions"].shape[1])]
    assert np.array_equal(opn.reward(actions, "Low").numpy(), fx["R_low"])
    assert np.array_equal(opn.reward(actions, "High").numpy(), fx["R_high"])
    assert fx["R_low"].tolist() == [0, 1, 2, 0, 0, 0]           # violated / dummy / boundary cases


def test_decision_margin_ignores_duplicate_rows():
    win = torch.tensor([[[1.0, 0.9999, 0.2]]])
    rows = torch.tensor([[[1., 2.], [1., 2.], [3., 4.]]])       # rows 0 and 1 identical
    assert abs(float(opn.decision_margin(win, rows)) - 0.8) < 1e-6
    rows2 = torch.tensor([[[1., 2.], [1., 3.], [3., 4.]]])
    assert abs(float(opn.decision_margin(win, rows2)) - 1e-4) < 1e-6


def _fx_data():
    with open(os.path.join(GOLDEN, "data_small.json")) as f:
        return json.load(f)


def test_data_oracle_reproduces_reference():
    fx = _fx_data()
    ds, K, T, P = fx["dataset"], fx["K"], fx["T"], fx["P"]
    assert odata.node_rows(ds["nodefeatures"]) == fx["loadData"]["nodefeatures"]
    assert odata.service_rows(ds["serviceFeature"]) == fx["loadData"]["serviceFeatureList"]
    ei, w = odata.service_graph(ds["labels"])
    assert ei == fx["loadData"]["edge_indices_service"] and w == fx["loadData"]["edge_attrs_service"]
    rows, labels = odata.load_data_pn(ds["nodefeatures"], ds["serviceFeature"], [fx["rank_shared"]] * P,
                                      ds["minCostList"], K)
    assert rows == fx["rows_shared"] and labels == ds["minCostList"]
    rows_e, _ = odata.load_data_pn(ds["nodefeatures"], ds["serviceFeature"], fx["rank_each"], ds["minCostList"], K)
    for p in range(P):
        for c in range(T):
            got = sorted({tuple(r) for r in rows_e[p][c * K:(c + 1) * K]})
            assert got == [tuple(r) for r in fx["rows_each_sets"][p][c]]
    k1, _ = odata.load_data_pn(ds["nodefeatures"], ds["serviceFeature"], [fx["rank_shared"]] * P, ds["minCostList"], 1)
    assert abs(odata.check(k1, ds["minCostList"], fx["check"]["actions"], T) - fx["check"]["score"]) < 1e-12


@pytest.mark.parametrize("name", ["tiny", "qws", "normal", "noservices"])
def test_ml_oracle_reproduces_reference_glue(name):
    fx = golden(f"ml_{name}.npz")
    torch.set_num_threads(1)
    sd = oml.make_state_dict(int(fx["hidden"]), int(fx["emb"]), int(fx["n_gin"]), int(fx["n_gcn"]), int(fx["seed"]) + 2)
    t = lambda k: torch.from_numpy(fx[k])   # noqa: E731
    data = oml.make_data(t("x"), t("edge_index"), t("batch"), t("x_service"), t("edge_index_service"),
                         t("edge_attr_service"))
    is_services = bool(fx["is_services"]) if "is_services" in fx else True      # modelML.py:151 / :157
    scores = oml.net_forward(sd, data, int(fx["n_gin"]), int(fx["n_gcn"]), is_services)
    assert float((scores - t("scores")).abs().max()) <= 1e-6
    assert torch.equal(oml.rank_services(t("scores")), t("ranking"))
    if not is_services:          # the fixture must really take the other branch
        assert float((oml.net_forward(sd, data, int(fx["n_gin"]), int(fx["n_gcn"]), True) - t("scores")).abs().max()) > 1e-3


def test_hand_graph():
    fx = golden("hand_graph.npz")
    x, ei, w = torch.from_numpy(fx["x"]), torch.from_numpy(fx["edge_index"]), torch.from_numpy(fx["w"])
    out = oml.gcn_conv(x, ei, w, torch.from_numpy(fx["weight"]), torch.from_numpy(fx["bias"]))
    assert torch.equal(out, torch.from_numpy(fx["gcn"]))
    # hand check of one entry: node 3 has in-edges {4->3 (w .125), self loop (w 1)}; deg = 1.125
    row, col, norm = oml.gcn_norm(ei, w, 5)
    sel = (col == 3)
    assert sorted(row[sel].tolist()) == [3, 4]
    deg3, deg4 = 0.125 + 1.0, 0.75 + 3.0
    assert abs(float(norm[sel & (row == 4)]) - 0.125 / (deg3 ** 0.5 * deg4 ** 0.5)) < 1e-7
    # existing self loops keep their weight (node 4: 3.0, node 2: 0.0 -> weight-0 loop)
    assert float(norm[(row == 2) & (col == 2)]) == 0.0


def test_rank_is_stable_lowest_index_first():
    s = torch.tensor([[0.5, 0.9, 0.5, 0.9, 0.1]])
    assert oml.rank_services(s).tolist() == [[1, 3, 0, 2, 4]]


def test_woa_oracle_reproduces_reference():
    """oracle/woa.py against the reference's own ESWOA runs (tests/golden/make_golden_woa.py): same draw count,
    same best-fitness history, same final selection — including the runs in which the reference's list aliasing
    and negative list positions change the outcome."""
    import copy
    from oracle import woa as owoa
    with open(os.path.join(GOLDEN, "woa_cases.json")) as f:
        cases = json.load(f)
    assert {c["name"] for c in cases} >= {"tiny", "no_solution", "foreign_pick", "violated", "qws_like", "normal_like"}
    for c in cases:
        got = owoa.eswoa([[tuple(s) for s in cat] for cat in c["services"]], c["constraints"], copy.deepcopy(c["solution"]),
                         c["pop_size"], c["max_iter"], owoa.DrawStream(c["seed"]))
        want = c["expected"]
        assert got["draws"] == want["draws"], c["name"]
        assert got["history"] == want["history"], c["name"]
        assert got["best_fitness"] == want["best_fitness"], c["name"]
        assert [list(r[:4]) for r in got["best_rows"]] == want["best_rows"], c["name"]
        assert [int(v) for v in got["best_pos"]] == want["best_pos"], c["name"]
        assert all(b <= a for a, b in zip(got["history"], got["history"][1:])), "best fitness never gets worse"


def test_woa_objective_known_answers():
    from oracle import woa as owoa
    rows = [(0.2, 0.9, 0.95, 0.98), (0.4, 0.5, 0.97, 0.99), (0.0, 1.0, 1.0, 1.0)]       # last: dummy row
    v, o = owoa.objective(rows, [[[0.9, 1.0]], [[0.99, 1.0]]])
    assert v == 1                                       # 0.95*0.97 = 0.9215 inside, 0.98*0.99 = 0.9702 below 0.99
    assert o == pytest.approx((0.6 / 2 + 1 - 0.5) / 2, abs=1e-15)
    s = owoa.DrawStream(7)
    u = [s.uniform() for _ in range(1000)]
    assert 0.0 <= min(u) and max(u) < 1.0 and abs(sum(u) / 1000 - 0.5) < 0.05 and s.count == 1000
    assert [owoa.DrawStream(7).below(10) for _ in range(3)] == [int(u[0] * 10)] * 3     # pure function of (seed, k)


@pytest.mark.parametrize("name", ["qws512", "normal1024"])
def test_pn_oracle_reproduces_reference_at_baseline_shape(name):
    """The BASELINE-size fixtures (tests/golden/make_golden_big.py: the real modelPN.py at QWS / Normal shape, inputs
    and weights regenerated from seeds): the oracle on the first chunk of 128 problems, whatever the thread count —
    identical picks up to the first fragile decision of every problem (tests/parity.py::prefix_parity)."""
    from parity import assert_R_parity, prefix_parity
    from pn_inputs import pn_inputs_chunked
    fx = golden(f"pn_big_{name}.npz")
    T, K, n = int(fx["n_cat"]), int(fx["n_per"]), 128 if name == "qws512" else 32
    x = pn_inputs_chunked(int(fx["B"]), T, K, int(fx["seed_inputs"]), int(fx["chunk"]))[:n]
    out = opn.two_level_greedy(opn.make_state_dict(256, int(fx["seed_low"])), opn.make_state_dict(256, int(fx["seed_high"])),
                               x, T, K)
    rec = prefix_parity(out["idx_low"], out["idx_high"], {k: fx[k][:n] for k in ("idx_low", "idx_high", "margin_low", "margin_high")},
                        name, x)
    assert rec["robust_identical"] == rec["robust_problems"] > 0
    assert_R_parity(out["R"], fx["R"][:n], name, rec["same_mask"])
    nw = min(n, fx["win_low"].shape[0])
    m = rec["same_mask"][:nw]
    assert np.allclose(out["win_low"].numpy()[:nw][m], fx["win_low"][:nw][m], rtol=0, atol=2e-5)


def test_ml_oracle_batched_forward_reproduces_reference_glue():
    """ml_pygbatch.npz: the reference's own Net.forward on a batch of two graphs assembled as torch_geometric 1.7.0's
    Data.__inc__ would (service edges of copy 1 shifted by graph 0's workflow node count).  The oracle's literal batched
    form reproduces it; the single-copy form does NOT (that is the documented divergence of the device pipeline)."""
    fx = golden("ml_pygbatch.npz")
    sd = oml.make_state_dict(int(fx["hidden"]), int(fx["emb"]), int(fx["n_gin"]), int(fx["n_gcn"]), int(fx["seed"]) + 2)
    t = lambda k: torch.from_numpy(fx[k])   # noqa: E731
    B, S = int(fx["B"]), int(fx["S"])
    ei, ea = oml.pyg_batch_service_edges(t("edge_index_service"), t("edge_attr_service"), fx["offsets"])
    data = oml.make_data(t("x"), t("edge_index"), t("batch"), t("x_service").repeat(B, 1), ei, ea)
    out = oml.net_forward_batched(sd, data, int(fx["n_gin"]), int(fx["n_gcn"]), S)
    assert float((out - t("scores")).abs().max()) <= 1e-6
    single = oml.net_forward(sd, oml.make_data(t("x"), t("edge_index"), t("batch"), t("x_service"),
                                               t("edge_index_service"), t("edge_attr_service")), int(fx["n_gin"]), int(fx["n_gcn"]))
    assert float((single - t("scores_single_copy")).abs().max()) <= 1e-6
    assert float((single - t("scores")).abs().max()) > 1e-3


SAMPLED_FORMS = ["bahdanau_g1_small", "dot_g2_small", "embed_small", "bahdanau_g1_qws", "embed_qws"]


@pytest.mark.parametrize("name", SAMPLED_FORMS)
def test_pn_oracle_sampled_forms_reproduce_reference(name):
    """pn_sample_<form>.npz: the real modelPN.py in sampling mode WITH 'Bahdanau' attention / glimpse rounds / the category
    embedding (modelPN.py:183-188,208-211,227-228; Tensor.multinomial routed to the counter-based stream).  The oracle's sampled
    forward of those forms reproduces picks, action_probs, action rows and R."""
    from parity import TAU_DRAW, assert_R_parity, prefix_parity
    fx = golden(f"pn_sample_{name}.npz")
    torch.set_num_threads(1)
    H, T, K, B, E = int(fx["hidden"]), int(fx["n_cat"]), int(fx["n_per"]), int(fx["B"]), int(fx["embedding_size"])
    att, ng = str(fx["attention"]), int(fx["n_glimpses"])
    x = torch.from_numpy(fx["inputs"])
    sds = [opn.make_state_dict(H, int(fx[k]), attention=att, embedding_size=E, n_cat=T) for k in ("seed_low", "seed_high")]
    out = opn.two_level_greedy(sds[0], sds[1], x, T, K, sample_high_seed=int(fx["sample_seed"]), attention=att, n_glimpses=ng)
    rec = prefix_parity(out["idx_low"], out["idx_high"], fx, name, x[:, :, -8:], tau_high=TAU_DRAW)
    assert rec["identical_problems"] >= B - 1
    s = rec["same_mask"]
    assert np.allclose(out["action_probs"].numpy()[s], fx["action_probs"][s], rtol=0, atol=1e-6)
    assert np.array_equal(out["actions"].numpy()[s], fx["actions"][s])
    assert_R_parity(out["R"], fx["R"], name, s)
    greedy = opn.two_level_greedy(sds[0], sds[1], x, T, K, attention=att, n_glimpses=ng)
    assert float((greedy["idx_high"].numpy() != fx["idx_high"]).mean()) > 0.3


@pytest.mark.parametrize("name", ["small", "qws"])
def test_pn_oracle_sampling_mode_reproduces_reference(name):
    """pn_sample_*.npz: the real modelPN.py in sampling mode (Low greedy -> latent, High sample='sample', the forward of a
    PNHigh training step) with Tensor.multinomial routed to the counter-based stream.  The oracle's sampled forward
    reproduces picks, action_probs and R; the draws really differ from the argmax."""
    from parity import TAU_DRAW, assert_R_parity, prefix_parity
    from pn_inputs import pn_inputs
    fx = golden(f"pn_sample_{name}.npz")
    torch.set_num_threads(1)
    H, T, K, B = int(fx["hidden"]), int(fx["n_cat"]), int(fx["n_per"]), int(fx["B"])
    x = pn_inputs(B, T, K, int(fx["seed_inputs"]))
    out = opn.two_level_greedy(opn.make_state_dict(H, int(fx["seed_low"])), opn.make_state_dict(H, int(fx["seed_high"])), x, T, K,
                               sample_high_seed=int(fx["sample_seed"]))
    rec = prefix_parity(out["idx_low"], out["idx_high"], fx, name, x, tau_high=TAU_DRAW)
    assert rec["identical_problems"] >= B - 1
    s = rec["same_mask"]
    assert np.allclose(out["action_probs"].numpy()[s], fx["action_probs"][s], rtol=0, atol=1e-6)
    assert_R_parity(out["R"], fx["R"], name, s)
    greedy = opn.two_level_greedy(opn.make_state_dict(H, int(fx["seed_low"])), opn.make_state_dict(H, int(fx["seed_high"])), x, T, K)
    assert float((greedy["idx_high"].numpy() != fx["idx_high"]).mean()) > 0.3
    u = [float(opn.stream_uniform24(7, c)) for c in range(4000)]
    assert 0.0 <= min(u) and max(u) < 1.0 and abs(np.mean(u) - 0.5) < 0.02      # the stream is a sane uniform


@pytest.mark.parametrize("name", ["small", "qws", "embed_small", "embed_qws", "bahdanau_g1_small", "dot_g2_small", "bahdanau_g0_small",
                                  "bahdanau_g1_qws"])
def test_training_oracle_reproduces_reference_autograd(name):
    """pn_train_*.npz: one REINFORCE step of the PNHigh trainer on the REAL reference modules and their autograd
    (trainPNHigh.py:83-108).  The oracle's restatement (oracle/pn_train.py) reproduces picks, loss, every gradient (2e-4
    relative per parameter; measured 4e-6) and the clipped-Adam weights (where |g| is clear of Adam's eps)."""
    from oracle import pn_train as optr
    from pn_inputs import pn_inputs
    fx = golden(f"pn_train_{name}.npz")
    torch.set_num_threads(4)
    H, T, K, B = int(fx["hidden"]), int(fx["n_cat"]), int(fx["n_per"]), int(fx["B"])
    E = int(fx["embedding_size"]) if "embedding_size" in fx.files else 0                  # "embed_*": embeddingTag=1 (round 5)
    x = pn_inputs(B, T, K, int(fx["seed_inputs"]))
    if E:
        x = torch.cat([torch.arange(T).repeat_interleave(K).float().view(1, T * K, 1).expand(B, T * K, 1), x], 2).contiguous()
    attention = str(fx["attention"]) if "attention" in fx.files else "Dot"               # 'Bahdanau' attention / glimpse rounds (round 5)
    n_glimpses = int(fx["n_glimpses"]) if "n_glimpses" in fx.files else 0
    sd_low, sd_high = (opn.make_state_dict(H, int(fx[k]), embedding_size=E, n_cat=T, attention=attention) for k in ("seed_low", "seed_high"))
    out = optr.train_step(sd_low, sd_high, x, T, K, int(fx["sample_seed"]), attention=attention, n_glimpses=n_glimpses)
    assert np.array_equal(out["idx_high"].numpy(), fx["idx_high"])
    assert abs(float(out["loss"]) - float(fx["loss"])) < 1e-6 and abs(float(out["grad_norm"]) - float(fx["grad_norm"])) < 1e-5
    g = torch.Generator().manual_seed(int(fx["seed_low"]))
    for k in optr.param_keys(sd_high):
        s = k.replace("actor.", "").replace(".", "_")
        got = out["grads"][k]
        if f"grad_{s}" in fx.files:
            want = torch.from_numpy(fx[f"grad_{s}"])
            assert float((got - want).norm() / (want.norm() + 1e-20)) < 2e-4, k
            clear = want.abs() > 1e-5
            assert float(((out["new_params"][k] - torch.from_numpy(fx[f"new_{s}"])).abs() * clear).max()) < 2e-6, k
        else:
            pos = torch.randint(0, got.numel(), (64,), generator=g)
            vec = torch.randn(got.numel(), generator=g)
            assert np.array_equal(pos.numpy(), fx[f"gradpos_{s}"])
            n = float(fx[f"gradnorm_{s}"])
            assert abs(float(got.norm()) - n) < 2e-4 * n, k
            assert float((got.flatten()[pos] - torch.from_numpy(fx[f"gradval_{s}"])).abs().max()) < 2e-4 * n + 1e-8, k
            assert abs(float((got.flatten().double() * vec.double()).sum()) - float(fx[f"gradproj_{s}"])) < 2e-3 * n + 1e-7, k


@pytest.mark.parametrize("name", ["tiny", "qws", "noservices"])
def test_ml_training_oracle_reproduces_reference_autograd(name):
    """ml_train_*.npz: TrainML.train's loop body (trainML.py:39-45) on the reference's own Net glue under autograd (stand-in
    convs); the autograd restatement oracle/ml_train.py gives the same loss, gradients, post-Adam weights and BatchNorm
    running statistics for the first step."""
    from oracle import ml_train as omt
    fx = golden(f"ml_train_{name}.npz")
    torch.set_num_threads(1)
    n_gin, n_gcn, S, B = int(fx["n_gin"]), int(fx["n_gcn"]), int(fx["S"]), int(fx["B"])
    sd = oml.make_state_dict(int(fx["hidden"]), int(fx["emb"]), n_gin, n_gcn, int(fx["seed"]) + 2)
    t = lambda k: torch.from_numpy(fx[k])   # noqa: E731
    ei, ea = oml.pyg_batch_service_edges(t("edge_index_service"), t("edge_attr_service"), fx["first_offsets"])
    data = oml.make_data(t("first_x"), t("first_edge_index"), t("first_batch"), t("x_service").repeat(B, 1), ei, ea)
    is_services = bool(int(fx["is_services"])) if "is_services" in fx.files else True      # "noservices": modelML.py:157-162
    out = omt.train_step(sd, data, t("first_y"), n_gin, n_gcn, S, float(fx["lr"]), is_services=is_services)
    assert abs(float(out["loss"]) - float(fx["first_loss"])) <= 1e-6
    assert float((out["scores"] - t("first_scores")).abs().max()) <= 1e-6
    keys = omt.trainable_keys(sd, n_gin, n_gcn, is_services)
    gmax = max(float(np.abs(fx[f"first_grad/{k}"]).max()) for k in keys)
    for k in keys:
        want = t(f"first_grad/{k}")
        assert float((out["grads"][k] - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-6 * gmax, k
        clear = want.abs() > 1e-6
        if bool(clear.any()):
            assert float((out["new_params"][k] - t(f"first_param/{k}"))[clear].abs().max()) <= 2e-6, k
    for pre, (rm, rv) in out["running"].items():
        assert float((rm - t(f"first_running_mean/{pre}")).abs().max()) <= 1e-5
        assert float((rv - t(f"first_running_var/{pre}")).abs().max()) <= 1e-5 * max(1.0, float(rv.abs().max()))


def test_gcn_conv_against_dense_fp64_formula():
    """Independent cross-check of the (unpinnable, third-party) GCNConv arithmetic: the layer written as the dense matrix
    formula  D^-1/2 (A_w + I) D^-1/2 X W + b  in float64 — weighted adjacency with A[dst, src] = w, unit self loops
    where none exist, degree = row sums over incoming weights — against oracle.ml.gcn_conv (the edge-list / scatter form
    restated from PyG 1.7.0).  Random weighted graph with a few self loops and duplicate edges."""
    g = torch.Generator().manual_seed(5)
    n, e, cin, cout = 60, 400, 12, 20
    ei = torch.randint(0, n, (2, e), generator=g)
    ei[:, :5] = torch.arange(5).repeat(2, 1)                       # explicit self loops keep their weight
    w = torch.rand(e, generator=g) + 0.1
    x = torch.randn(n, cin, generator=g)
    W = torch.randn(cin, cout, generator=g) * 0.3
    b = torch.randn(cout, generator=g)
    A = torch.zeros(n, n, dtype=torch.float64)
    for k in range(e):
        s, d = int(ei[0, k]), int(ei[1, k])
        if s != d:
            A[d, s] += float(w[k])                                  # duplicates add up, as scatter_add does
    loop = torch.ones(n, dtype=torch.float64)
    for k in range(e):
        if int(ei[0, k]) == int(ei[1, k]):
            loop[int(ei[0, k])] = float(w[k])                       # add_remaining_self_loops: an existing loop keeps its weight
    A += torch.diag(loop)
    deg = A.sum(1)
    dis = deg.pow(-0.5)
    want = (dis[:, None] * A * dis[None, :]) @ (x.double() @ W.double()) + b.double()
    got = oml.gcn_conv(x, ei, w, W, b)
    assert float((got.double() - want).abs().max()) < 2e-5


def test_committed_fixtures_are_what_the_generators_wrote():
    """Every fixture in tests/golden is byte for byte (sha256) the file its generator wrote, with the key list the generator
    writes today: MANIFEST.json is written by the generators themselves (tests/golden/manifest.py).  A fixture regenerated and not
    committed, or committed from an older generator, fails here instead of being papered over by a default in a test
    (VERDICT r5, item 4).  No fixture may be missing from the manifest and no manifest entry from the directory."""
    import sys
    sys.path.insert(0, GOLDEN)
    import manifest
    with open(manifest.PATH) as f:
        want = json.load(f)
    have = manifest.fixture_files()
    assert sorted(want) == have, (sorted(set(want) ^ set(have)))
    for name in have:
        got = manifest.describe(name)
        assert got == want[name], f"{name}: committed file differs from what its generator wrote ({got} != {want[name]})"
    # the REINFORCE fixtures all say which attention form they were made with (the four stale ones of round 5 did not)
    for name in have:
        if name.startswith("pn_train_"):
            assert {"attention", "n_glimpses"} <= set(want[name]["keys"]), name
