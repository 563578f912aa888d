"""-m gpu: the REINFORCE training step of the High-level pointer network (reference
src/models/trainPNHigh.py:76-112) — hand-written backward, clipping and Adam — against fixtures produced by the REAL
reference modules and their autograd (tests/golden/make_golden.py::gen_pn_train) and against the live autograd oracle."""
import numpy as np
import pytest
import torch

from conftest import golden, record_agreement
from oracle import pn as opn
from oracle import pn_train as optr
from pn_inputs import pn_inputs

pytestmark = pytest.mark.gpu
GRAD_RTOL = 2e-4        # per parameter: ||g - g_ref|| / ||g_ref||
# 'Bahdanau' attention + a glimpse round at the QWS shape: the step's advantages nearly cancel (loss -4e-4, gradient norm 0.15), so
# the fp32 summation order of 47 steps x 235 positions shows — the autograd oracle itself stands at 1.4e-4 from the reference here
GRAD_RTOL_OF = {"bahdanau_g1_qws": 5e-4}


def _E(fx):
    try:
        return int(fx["embedding_size"])
    except (KeyError, ValueError):
        return 0


def _form(fx):
    """(attention, n_glimpses) of a fixture: every pn_train_* FILE carries both (tests/golden/MANIFEST.json pins the key lists); a
    live configuration (a plain dict, no fixture) that names neither is the shipped ('Dot', 0)."""
    if isinstance(fx, dict) and "attention" not in fx:
        return "Dot", 0
    return str(fx["attention"]), int(fx["n_glimpses"])


def _nets(fx, dev):
    from gnnpn_sc_amd.modelPN import CombinatorialRL, reward
    H, T, K, E = int(fx["hidden"]), int(fx["n_cat"]), int(fx["n_per"]), _E(fx)
    attention, n_glimpses = _form(fx)
    nets = []
    for level, seed in (("Low", int(fx["seed_low"])), ("High", int(fx["seed_high"]))):
        m = CombinatorialRL(E, H, T * K, n_glimpses, 10, 1, reward, attention, K, T, use_cuda=True, level=level)
        m.load_state_dict(opn.make_state_dict(H, seed, embedding_size=E, n_cat=T, attention=attention), strict=True)
        nets.append(m.to(dev))
    return nets


def _inputs(fx):
    """The fixture's PN inputs; with the category embedding (embeddingTag=1) column 0 is the category (loadData.py:130-148)."""
    T, K, B = int(fx["n_cat"]), int(fx["n_per"]), int(fx["B"])
    x = pn_inputs(B, T, K, int(fx["seed_inputs"]))
    if _E(fx):
        cat = torch.arange(T).repeat_interleave(K).float().view(1, T * K, 1).expand(B, T * K, 1)
        x = torch.cat([cat, x], 2).contiguous()
    return x


def _short(k):
    return k.replace("actor.", "").replace(".", "_")


FORMS = ["bahdanau_g1_small", "dot_g2_small", "bahdanau_g0_small", "bahdanau_g1_qws"]      # the switched-off attention forms (round 5)


@pytest.mark.parametrize("name", ["small", "qws", "embed_small", "embed_qws"] + FORMS)
def test_actor_gradients_golden(dev, name):
    """Backward with the fixture's picks given (teacher forcing): every actor gradient against the reference's autograd —
    in full at H = 32, as norm, 64 seeded entries and a seeded projection per parameter at H = 256 — then clip + Adam.
    ("embed_*": embeddingTag=1 — the category embedding in front of embedding2, its table trained too; "bahdanau_*" / "dot_g2":
    'Bahdanau' attention and / or glimpse rounds, the two Attention modules' parameters trained too; round 5.)"""
    from gnnpn_sc_amd import ops
    from gnnpn_sc_amd.trainPNHigh import ActorAdam, actor_gradients
    fx = golden(f"pn_train_{name}.npz")
    low, high = _nets(fx, dev)
    T, K, B = int(fx["n_cat"]), int(fx["n_per"]), int(fx["B"])
    x = _inputs(fx).to(dev)
    KEYS = optr.param_keys(opn.make_state_dict(int(fx["hidden"]), int(fx["seed_high"]), embedding_size=_E(fx), n_cat=T, attention=_form(fx)[0]))
    R = torch.from_numpy(fx["R"]).to(dev)
    gscale = ((R - R.mean()) / B).contiguous()                     # first batch: critic = R.mean() (trainPNHigh.py:87-92)
    idx = torch.from_numpy(fx["idx_high"]).int().to(dev)
    win_low = torch.from_numpy(fx["win_low"]).to(dev)
    grads, logp = actor_gradients(high.actor, x, idx, win_low, gscale)
    ops.check_status(dev)
    loss = float(((R - R.mean()) * logp.sum(1)).mean())
    assert abs(loss - float(fx["loss"])) < 2e-5 * max(1.0, abs(float(fx["loss"]))), (loss, float(fx["loss"]))
    worst, rtol = 0.0, GRAD_RTOL_OF.get(name, GRAD_RTOL)
    g = torch.Generator().manual_seed(int(fx["seed_low"]))         # the generator's seed = `seed` of gen_pn_train
    assert set(grads) == {k.replace("actor.", "") for k in KEYS}
    for k in KEYS:
        got = grads[k.replace("actor.", "")].detach().cpu()
        s = _short(k)
        if f"grad_{s}" in fx.files:
            want = torch.from_numpy(fx[f"grad_{s}"])
            rel = float((got - want).norm() / (want.norm() + 1e-20))
        else:
            flat = got.flatten()
            pos = torch.randint(0, flat.numel(), (64,), generator=g)          # the generator's draws, in its order
            vec = torch.randn(flat.numel(), generator=g)
            assert np.array_equal(pos.numpy(), fx[f"gradpos_{s}"])
            want_n = float(fx[f"gradnorm_{s}"])
            rel = abs(float(flat.norm()) - want_n) / (want_n + 1e-20)
            assert float((flat[pos] - torch.from_numpy(fx[f"gradval_{s}"])).abs().max()) < rtol * want_n + 1e-7, k
            proj = float((flat.double() * vec.double()).sum())                # a seeded random projection of the whole tensor
            assert abs(proj - float(fx[f"gradproj_{s}"])) < 2 * rtol * want_n * float(vec.norm()) / np.sqrt(flat.numel()) * 4 + 1e-7, k
        worst = max(worst, rel)
        assert rel < rtol, f"{k}: relative gradient error {rel:.2e}"
    record_agreement(f"train_gradients_{name}", {"worst_relative_gradient_error": worst, "loss": loss, "loss_reference": float(fx["loss"]),
                                                 "parameters": len(KEYS), "B": B, "T": T, "K": K})
    # clip_grad_norm_ + Adam (first step)
    opt = ActorAdam(high.actor, lr=0.5e-4, max_grad_norm=2.0)
    norm = float(opt.step(grads))
    assert abs(norm - float(fx["grad_norm"])) < 2e-4 * float(fx["grad_norm"])
    for k in KEYS:
        s = _short(k)
        new = dict(high.actor.named_parameters())[k.replace("actor.", "")].detach().cpu()
        if f"new_{s}" in fx.files:
            want, gref = torch.from_numpy(fx[f"new_{s}"]), torch.from_numpy(fx[f"grad_{s}"])
            clear = gref.abs() > 1e-5                               # Adam's first step is ill-conditioned near g = 0
            assert float(((new - want).abs() * clear).max()) < 3e-6, k
            assert float((new - want).abs().max()) <= 1.01e-4, k
        else:
            pos = torch.from_numpy(fx[f"gradpos_{s}"])
            clear = torch.from_numpy(np.abs(fx[f"gradval_{s}"]) > 1e-5)
            assert float(((new.flatten()[pos] - torch.from_numpy(fx[f"newval_{s}"])).abs() * clear).max()) < 3e-6, k


def test_actor_gradients_vs_live_autograd(dev):
    """A shape that is not a stored fixture (T=9, K=4, B=12, H=256, one dummy category): gradients against the oracle's
    torch-autograd restatement run live, including the seeded projection check on the big matrices."""
    from gnnpn_sc_amd.trainPNHigh import actor_gradients
    cfg = {"hidden": 256, "n_cat": 9, "n_per": 4, "seed_low": 301, "seed_high": 302}
    low, high = _nets(cfg, dev)
    T, K, B = 9, 4, 12
    x = pn_inputs(B, T, K, 303, dummy_every=4)
    sd_low, sd_high = opn.make_state_dict(256, 301), opn.make_state_dict(256, 302)
    ref = optr.train_step(sd_low, sd_high, x, T, K, sample_seed=99)
    gscale = (ref["advantage"] / B).contiguous().to(dev)
    grads, logp = actor_gradients(high.actor, x.to(dev), ref["idx_high"].int().to(dev), ref["win_low"].to(dev), gscale)
    assert float((logp.cpu().exp() - ref["pick_prob"]).abs().max()) < 1e-5
    for k in optr.PARAM_KEYS:
        got, want = grads[k.replace("actor.", "")].cpu(), ref["grads"][k]
        rel = float((got - want).norm() / (want.norm() + 1e-20))
        assert rel < GRAD_RTOL, f"{k}: {rel:.2e}"


@pytest.mark.parametrize("name", ["small", "qws", "embed_small", "embed_qws"] + FORMS)
def test_train_step_end_to_end(dev, name):
    """TrainModel.train_step (trainPNHigh.py:81-110 in one call): sampled forward with the fixture's stream, backward, clip,
    Adam.  Where the drawn picks equal the reference's (they do unless a draw is fragile) loss and gradient norm match, the
    weights move, the inference kernels see the new weights, and a second step runs from the updated critic."""
    from gnnpn_sc_amd.modelPN import two_level_greedy
    from gnnpn_sc_amd.trainPNHigh import TrainModel
    fx = golden(f"pn_train_{name}.npz")
    low, high = _nets(fx, dev)
    T, K, B = int(fx["n_cat"]), int(fx["n_per"]), int(fx["B"])
    x = _inputs(fx)
    tm = TrainModel(high, None, None, 1, 0.9, True, "QWS", T, lr=0.5e-4, batch_size=B, max_grad_norm=2., low_model=low, device=str(dev))
    before = {k: p.detach().clone() for k, p in high.actor.named_parameters()}
    greedy_before = two_level_greedy(low, high, x.to(dev))
    out = tm.train_step(x, sample_seed=int(fx["sample_seed"]))
    same = np.array_equal(out["idx_high"].cpu().numpy(), fx["idx_high"])
    if same:
        assert abs(float(out["loss"]) - float(fx["loss"])) < 2e-5 * max(1.0, abs(float(fx["loss"])))
        assert abs(float(out["grad_norm"]) - float(fx["grad_norm"])) < 2e-4 * float(fx["grad_norm"])
    assert float((out["R"].cpu() - torch.from_numpy(fx["R"])).abs().max()) < 1e-4 or not same
    moved = max(float((p.detach() - before[k]).abs().max()) for k, p in high.actor.named_parameters())
    assert 1e-5 < moved <= 1.01e-4                                  # Adam's first step: lr-sized moves (lr = 0.5e-4 ... bias-corrected)
    greedy_after = two_level_greedy(low, high, x.to(dev))           # the inference path re-packs the updated weights
    assert not torch.equal(greedy_after["win_high_raw"], greedy_before["win_high_raw"])
    out2 = tm.train_step(x)                                         # critic moving average (:89-90), own sampling stream
    assert np.isfinite(float(out2["loss"])) and tm.actor_optim.steps == 2 and len(tm.train_tour) == 2


def test_training_drivers_end_to_end(dev, tmp_path, monkeypatch):
    """`PNLow(...).start()` then `PNHigh(...).start()` (trainPNLow.py:172-223, trainPNHigh.py:175-251) on a dataset in the
    reference's JSON formats: the Low net trains and leaves the checkpoint / allActions / allR / val artefacts, the High
    net loads that checkpoint (trainPNHigh.py:237-242), trains against it and leaves checkpoints and the allActions file
    that `ML2PN.check(dataset, T, epoch)` scores; weights move, rewards stay finite."""
    import contextlib
    import io
    import json
    import os
    import gnnpn_sc_amd.synth as synth
    from conftest import GOLDEN
    from gnnpn_sc_amd import ML2PN
    from gnnpn_sc_amd.trainPNHigh import PNHigh, PNLow
    with open(os.path.join(GOLDEN, "data_small.json")) as f:
        fx = json.load(f)
    T, K, P = fx["T"], fx["K"], fx["P"]
    synth.write_dataset(str(tmp_path), "QWS", fx["dataset"])
    monkeypatch.chdir(tmp_path)
    os.makedirs("solutions/pretrained")
    with open("solutions/pretrained/QWS-ML.txt", "w") as f:
        json.dump(fx["rank_each"], f)
    low = PNLow("QWS", 0, 1, T, 1, K, 256, 0, 10, 1, 0.9, 2.0, 1e-4, -1).start(n_epochs=2, device=str(dev), batch_size=4)
    assert low.actor_optim.steps == 2 * 3 and len(low.train_tour) == 6              # 12 training problems / 4 per batch
    for name in ("epoch0.model", "epoch1.model", "allActions1.txt", "allR1.txt", "val1.txt"):
        assert os.path.exists(f"solutions/PNLow/QWS/{name}"), name
    with open("solutions/PNLow/QWS/allActions1.txt") as f:
        acts = json.load(f)
    assert len(acts) == T + 2 and len(acts[0]) == P // 4 and len(acts[0][0]) == 8  # trainPNLow.py:122 sizes the list T + 2
    high = PNHigh("QWS", 0, 1, T, 1, K, 256, 0, 10, 1, 0.9, 2.0, 0.5e-4, -1, 1).start(n_epochs=2, device=str(dev), batch_size=4)
    assert high.actor_optim.steps == 6 and all(np.isfinite(v) for v in high.train_tour)
    sd = torch.load("solutions/PNHigh/QWS/epoch1.model", map_location="cpu")["model"]
    assert set(sd) == set(torch.load("solutions/PNLow/QWS/epoch1.model", map_location="cpu")["model"])
    low_ck = torch.load("solutions/PNHigh/QWS/epoch1_low.model", map_location="cpu")["model"]
    assert all(torch.equal(low_ck[k], torch.load("solutions/PNLow/QWS/epoch1.model", map_location="cpu")["model"][k]) for k in low_ck)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        score = ML2PN.check("QWS", T, 1)                                            # ./solutions/PNHigh/QWS/allActions1.txt
    assert buf.getvalue().split()[0] == "1" and np.isfinite(score)
    # embeddingTag=1 (trainPNLow.py:190-199, trainPNHigh.py:197-204; switched off by the shipped configs): rows keep their
    # category column, embedding_size = 20, the category table is trained with the rest (round 5: used to raise)
    low_e = PNLow("QWS", 1, 1, T, 1, K, 256, 0, 10, 1, 0.9, 2.0, 1e-4, -1).start(n_epochs=1, device=str(dev), batch_size=4)
    e1 = low_e.model.actor.embedding1.weight.detach().cpu()
    assert low_e.actor_optim.steps == 3 and tuple(e1.shape) == (T, 20) and all(np.isfinite(v) for v in low_e.train_tour)
    # its artefacts live under <ds>/20embeddings/ (trainPNLow.py:190-191: self.dataset += "20embeddings/"), beside — not over — the
    # embeddingTag=0 run's, and PNHigh(embeddingTag=1) finds its Low net there (trainPNHigh.py:197-198,237-240)
    ck = torch.load("solutions/PNLow/QWS/20embeddings/epoch0.model", map_location="cpu")["model"]
    assert torch.equal(ck["actor.embedding1.weight"], e1)
    assert "actor.embedding1.weight" not in torch.load("solutions/PNLow/QWS/epoch0.model", map_location="cpu")["model"]   # untouched
    high_e = PNHigh("QWS", 1, 1, T, 1, K, 256, 0, 10, 1, 0.9, 2.0, 0.5e-4, -1, 0).start(n_epochs=1, device=str(dev), batch_size=4)
    assert high_e.actor_optim.steps == 3 and all(np.isfinite(v) for v in high_e.train_tour)
    with open("solutions/PNHigh/QWS/allActions0.txt") as f:
        assert len(json.load(f)[0][0]) == 8                                         # the embeddingTag=0 run's file: still 8 columns
    with open("solutions/PNHigh/QWS/20embeddings/allActions0.txt") as f:
        acts_e = json.load(f)
    assert len(acts_e) == T and len(acts_e[0][0]) == 9                              # action rows WITH their category column
    # n_glimpses = 1 (environment.ini ships 0; the drivers pass it through, attention stays 'Dot': trainPNLow.py:205-210,
    # trainPNHigh.py:207-231): sampled forward and REINFORCE step through a glimpse round (round 5: used to raise)
    low_g = PNLow("QWS", 0, 1, T, 1, K, 256, 1, 10, 1, 0.9, 2.0, 1e-4, -1).start(n_epochs=1, device=str(dev), batch_size=4)
    assert low_g.actor_optim.steps == 3 and all(np.isfinite(v) for v in low_g.train_tour)
    high_g = PNHigh("QWS", 0, 1, T, 1, K, 256, 1, 10, 1, 0.9, 2.0, 0.5e-4, -1, 0).start(n_epochs=1, device=str(dev), batch_size=4)
    assert high_g.actor_optim.steps == 3 and all(np.isfinite(v) for v in high_g.train_tour)
