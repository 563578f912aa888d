#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
The reference's Python is imported from where it lies — nothing of it is copied into the repo;
the fixtures hold inputs and expected outputs only.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz, *.json

What it pins:
  pn_*.npz      two-level greedy decode of the real ``src/models/modelPN.py`` (Tensor.cuda no-op'd)
  reward.npz    ``reward``/``calc`` known-answer cases (dummy rows, violated constraints)
  data_small.json  ``loadData``, ``loadDataPN``, ``ML2PN.check`` of the real ``src/loadData.py`` /
                ``src/ML2PN.py`` on a synthetic dataset written in the reference's JSON formats
  ml_*.npz      the real ``Net.__init__``/``forward`` glue of ``src/models/modelML.py`` with the
                third-party convs stood in (tests/golden/pyg_standin.py)
Each section also asserts that the oracle (oracle/*.py) reproduces the reference output, so a
fixture is never written from an oracle that disagrees with the reference.
"""
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import gnnpn_sc_amd.synth as synth          # noqa: E402
from oracle import data as odata            # noqa: E402
from oracle import ml as oml                # noqa: E402
from oracle import pn as opn                # noqa: E402
import pyg_standin                          # noqa: E402
from pn_inputs import pn_inputs             # noqa: E402,F401


def import_reference():
    # the reference hard-codes CUDA (modelPN.py:151,198; modelML.py:171): make those no-ops
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    _to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple(torch.device("cpu") if isinstance(x, torch.device) and x.type == "cuda" else x for x in a)
        return _to(self, *a, **k)
    torch.Tensor.to = to
    pyg_standin.install()
    sys.path.insert(0, REF)
    from src.models import modelPN, modelML
    from src import loadData, ML2PN
    return modelPN, modelML, loadData, ML2PN


def gen_pn(modelPN, name, H, T, K, B, seed, dummy_every=0, weight_scale=1.0):
    sd_low, sd_high = opn.make_state_dict(H, seed), opn.make_state_dict(H, seed + 1)
    if weight_scale != 1.0:
        # large recurrent weights drive |dot| past ~9 where 10*tanh(x) rounds to exactly 10.0f: exact ties
        # inside a window, which torch.max resolves to the FIRST maximum (SURVEY.md §7 "hard parts" (i))
        for sd in (sd_low, sd_high):
            for k in sd:
                if "encoder" in k or "decoder." in k:
                    sd[k] = sd[k] * weight_scale
    L = T * K

    def build(level, sd):
        m = modelPN.CombinatorialRL(0, H, L, 0, 10, 1, modelPN.reward, "Dot", K, T,
                                    use_cuda=False, level=level)
        m.load_state_dict(sd, strict=True)
        return m.eval()

    low, high = build("Low", sd_low), build("High", sd_high)
    x = pn_inputs(B, T, K, seed + 2, dummy_every)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        # exactly trainPNHigh.py:138-139
        _, _, _, idx_low, latent = low(x, None, sample="greedy", training="SL")
        R, probs, actions, idx_high, logits_high = high(x, None, latent, sample="greedy")
        R_low, _, _, _, _ = low(x, None, sample="greedy", training="RL")
    ref = {
        "idx_low": torch.stack(idx_low, 1), "idx_high": torch.stack(idx_high, 1), "R": R, "R_low": R_low,
        "actions": torch.stack(actions, 1), "action_probs": torch.stack(probs, 1),
        "win_low": torch.stack([latent[k][:, k * K:(k + 1) * K] for k in range(T)], 1),
        "win_high": torch.stack([(logits_high[k] + latent[k])[:, k * K:(k + 1) * K] for k in range(T)], 1),
    }
    orc = opn.two_level_greedy(sd_low, sd_high, x, T, K)
    for key in ("idx_low", "idx_high", "R", "actions", "action_probs", "win_low", "win_high"):
        assert torch.equal(ref[key], orc[key]), f"oracle != reference on {key} ({name})"
    # full-length latent of the Low net keeps -inf at previously chosen positions (modelPN.py:172)
    lat1 = latent[1]
    assert torch.isinf(lat1[torch.arange(B), idx_low[0]]).all()
    out = {k: v.numpy() for k, v in ref.items()}
    out.update(inputs=x.numpy(), hidden=H, n_cat=T, n_per=K, seed_low=seed, seed_high=seed + 1, weight_scale=weight_scale,
               margin_low=orc["margin_low"].numpy(), margin_high=orc["margin_high"].numpy(),
               latent_step1=lat1.numpy())
    np.savez_compressed(os.path.join(HERE, f"pn_{name}.npz"), **out)
    n_tie = int((ref["win_low"] == 10.0).sum() + (ref["win_low"] == -10.0).sum())
    print(f"pn_{name}: B={B} T={T} K={K} H={H} saturated window logits: {n_tie}; min margin low/high = "
          f"{float(orc['margin_low'].min()):.3e} / {float(orc['margin_high'].min()):.3e}")


def gen_pn_attn(modelPN, name, H, T, K, B, seed, attention, n_glimpses):
    """the attention forms the reference's configs leave switched off — 'Bahdanau' attention
    (modelPN.py:80-90,103-109) and glimpse rounds (:208-211) — run through the REAL modules, Low ("SL") -> latent -> High."""
    sd_low = opn.make_state_dict(H, seed, attention=attention)
    sd_high = opn.make_state_dict(H, seed + 1, attention=attention)
    L = T * K

    def build(level, sd):
        m = modelPN.CombinatorialRL(0, H, L, n_glimpses, 10, 1, modelPN.reward, attention, K, T, use_cuda=False, level=level)
        m.load_state_dict(sd, strict=True)
        return m.eval()

    low, high = build("Low", sd_low), build("High", sd_high)
    x = pn_inputs(B, T, K, seed + 2, 0)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        _, _, _, idx_low, latent = low(x, None, sample="greedy", training="SL")
        R, probs, actions, idx_high, logits_high = high(x, None, latent, sample="greedy")
    ref = {
        "idx_low": torch.stack(idx_low, 1), "idx_high": torch.stack(idx_high, 1), "R": R,
        "actions": torch.stack(actions, 1), "action_probs": torch.stack(probs, 1),
        "win_low": torch.stack([latent[k][:, k * K:(k + 1) * K] for k in range(T)], 1),
        "win_high": torch.stack([(logits_high[k] + latent[k])[:, k * K:(k + 1) * K] for k in range(T)], 1),
    }
    orc = opn.two_level_greedy(sd_low, sd_high, x, T, K, attention=attention, n_glimpses=n_glimpses)
    for key in ("idx_low", "idx_high", "R", "actions", "action_probs", "win_low", "win_high"):
        assert torch.equal(ref[key], orc[key]), f"oracle != reference on {key} ({name})"
    out = {k: v.numpy() for k, v in ref.items()}
    out.update(inputs=x.numpy(), hidden=H, n_cat=T, n_per=K, seed_low=seed, seed_high=seed + 1, attention=attention,
               n_glimpses=n_glimpses, margin_low=orc["margin_low"].numpy(), margin_high=orc["margin_high"].numpy())
    np.savez_compressed(os.path.join(HERE, f"pn_attn_{name}.npz"), **out)
    print(f"pn_attn_{name}: {attention} glimpses={n_glimpses} B={B} T={T} K={K} H={H}; min margin low/high = "
          f"{float(orc['margin_low'].min()):.3e} / {float(orc['margin_high'].min()):.3e}")


def gen_pn_embed(modelPN, name, H, T, K, B, seed, E):
    """embedding_size != 0 (embeddingTag=1; modelPN.py:153-154, 183-188, reward's tag :42-45): rows [category | 8 floats], the
    category embedded and concatenated in front of embedding2 — run through the REAL modules, Low ("SL") -> latent -> High."""
    sd_low = opn.make_state_dict(H, seed, embedding_size=E, n_cat=T)
    sd_high = opn.make_state_dict(H, seed + 1, embedding_size=E, n_cat=T)
    L = T * K

    def build(level, sd):
        m = modelPN.CombinatorialRL(E, H, L, 0, 10, 1, modelPN.reward, "Dot", K, T, use_cuda=False, level=level)
        m.load_state_dict(sd, strict=True)
        return m.eval()

    low, high = build("Low", sd_low), build("High", sd_high)
    x8 = pn_inputs(B, T, K, seed + 2, 0)
    cat = torch.arange(T).repeat_interleave(K).float().view(1, L, 1).expand(B, L, 1)     # loadData.py:130-148: column 0 = the category
    x = torch.cat([cat, x8], 2).contiguous()
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        _, _, _, idx_low, latent = low(x, None, sample="greedy", training="SL")
        R, probs, actions, idx_high, logits_high = high(x, None, latent, sample="greedy")
    ref = {
        "idx_low": torch.stack(idx_low, 1), "idx_high": torch.stack(idx_high, 1), "R": R,
        "actions": torch.stack(actions, 1), "action_probs": torch.stack(probs, 1),
        "win_low": torch.stack([latent[k][:, k * K:(k + 1) * K] for k in range(T)], 1),
        "win_high": torch.stack([(logits_high[k] + latent[k])[:, k * K:(k + 1) * K] for k in range(T)], 1),
    }
    orc = opn.two_level_greedy(sd_low, sd_high, x, T, K)
    for key in ("idx_low", "idx_high", "R", "actions", "action_probs", "win_low", "win_high"):
        assert torch.equal(ref[key], orc[key]), f"oracle != reference on {key} ({name})"
    out = {k: v.numpy() for k, v in ref.items()}
    out.update(inputs=x.numpy(), hidden=H, n_cat=T, n_per=K, seed_low=seed, seed_high=seed + 1, embedding_size=E,
               margin_low=orc["margin_low"].numpy(), margin_high=orc["margin_high"].numpy())
    np.savez_compressed(os.path.join(HERE, f"pn_embed_{name}.npz"), **out)
    print(f"pn_embed_{name}: E={E} B={B} T={T} K={K} H={H}; min margin low/high = "
          f"{float(orc['margin_low'].min()):.3e} / {float(orc['margin_high'].min()):.3e}")


def gen_pn_sample(modelPN, name, H, T, K, B, seed, sample_seed):
    """The sampling mode (modelPN.py:227-228) of the REAL reference — Low greedy -> latent, High with sample='sample', the
    forward of a PNHigh training step (trainPNHigh.py:83-84) — with ``Tensor.multinomial`` routed to the counter-based
    stream (oracle.pn.multinomial_from_stream), which makes the run a function of (weights, inputs, seed).  Also checks
    that the reference's re-draw branch (:229-234) never fires (it prints ' RESAMPLE!')."""
    sd_low, sd_high = opn.make_state_dict(H, seed), opn.make_state_dict(H, seed + 1)
    L = T * K

    def build(level, sd):
        m = modelPN.CombinatorialRL(0, H, L, 0, 10, 1, modelPN.reward, "Dot", K, T, use_cuda=False, level=level)
        m.load_state_dict(sd, strict=True)
        return m.eval()

    low, high = build("Low", sd_low), build("High", sd_high)
    x = pn_inputs(B, T, K, seed + 2)
    state = {"k": 0}
    real = torch.Tensor.multinomial

    def routed(self, num_samples=1, replacement=False, generator=None):
        assert num_samples == 1
        idx, _ = opn.multinomial_from_stream(self, state["k"], T, K, sample_seed)
        state["k"] += 1
        return idx.view(-1, 1)

    torch.Tensor.multinomial = routed
    buf = io.StringIO()
    try:
        with torch.no_grad(), contextlib.redirect_stdout(buf):
            _, _, _, idx_low, latent = low(x, None, sample="greedy", training="SL")
            R, probs, actions, idx_high, _ = high(x, None, latent)            # default sample="sample", training="RL"
    finally:
        torch.Tensor.multinomial = real
    assert state["k"] == T and "RESAMPLE" not in buf.getvalue()
    ref = {"idx_low": torch.stack(idx_low, 1), "idx_high": torch.stack(idx_high, 1), "R": R,
           "actions": torch.stack(actions, 1), "action_probs": torch.stack(probs, 1)}
    orc = opn.two_level_greedy(sd_low, sd_high, x, T, K, sample_high_seed=sample_seed)
    for key in ref:
        assert torch.equal(ref[key], orc[key]), f"oracle != reference on {key} ({name})"
    greedy = opn.two_level_greedy(sd_low, sd_high, x, T, K)
    frac = float((greedy["idx_high"] != ref["idx_high"]).float().mean())
    out = {k: v.numpy() for k, v in ref.items()}
    out.update(hidden=H, n_cat=T, n_per=K, B=B, seed_low=seed, seed_high=seed + 1, seed_inputs=seed + 2,
               sample_seed=sample_seed, margin_low=orc["margin_low"].numpy(), margin_high=orc["margin_high"].numpy())
    np.savez_compressed(os.path.join(HERE, f"pn_sample_{name}.npz"), **out)
    print(f"pn_sample_{name}: B={B} T={T} K={K} H={H}: {frac:.2f} of the sampled picks differ from the greedy ones; "
          f"min draw margin {float(orc['margin_high'].min()):.2e}")


def gen_pn_sample_forms(modelPN, name, H, T, K, B, seed, sample_seed, attention="Dot", n_glimpses=0, E=0):
    """The sampling mode (modelPN.py:227-228) TOGETHER with the forms the shipped configurations leave switched off —
    'Bahdanau' attention (:80-90,103-109), glimpse rounds (:208-211), the category embedding (:153-154,183-188) — through the
    REAL modules: Low greedy -> latent, High with sample='sample', ``Tensor.multinomial`` routed to the counter-based stream as
    in gen_pn_sample.  Inputs are stored (the embedding form's rows carry the category column)."""
    sd_low = opn.make_state_dict(H, seed, attention=attention, embedding_size=E, n_cat=T)
    sd_high = opn.make_state_dict(H, seed + 1, attention=attention, embedding_size=E, n_cat=T)
    L = T * K

    def build(level, sd):
        m = modelPN.CombinatorialRL(E, H, L, n_glimpses, 10, 1, modelPN.reward, attention, K, T, use_cuda=False, level=level)
        m.load_state_dict(sd, strict=True)
        return m.eval()

    low, high = build("Low", sd_low), build("High", sd_high)
    x = pn_inputs(B, T, K, seed + 2, 0)
    if E:
        cat = torch.arange(T).repeat_interleave(K).float().view(1, L, 1).expand(B, L, 1)     # loadData.py:130-148: column 0 = the category
        x = torch.cat([cat, x], 2).contiguous()
    state = {"k": 0}
    real = torch.Tensor.multinomial

    def routed(self, num_samples=1, replacement=False, generator=None):
        assert num_samples == 1
        idx, _ = opn.multinomial_from_stream(self, state["k"], T, K, sample_seed)
        state["k"] += 1
        return idx.view(-1, 1)

    torch.Tensor.multinomial = routed
    buf = io.StringIO()
    try:
        with torch.no_grad(), contextlib.redirect_stdout(buf):
            _, _, _, idx_low, latent = low(x, None, sample="greedy", training="SL")
            R, probs, actions, idx_high, _ = high(x, None, latent)            # default sample="sample", training="RL"
    finally:
        torch.Tensor.multinomial = real
    assert state["k"] == T and "RESAMPLE" not in buf.getvalue()
    ref = {"idx_low": torch.stack(idx_low, 1), "idx_high": torch.stack(idx_high, 1), "R": R,
           "actions": torch.stack(actions, 1), "action_probs": torch.stack(probs, 1)}
    orc = opn.two_level_greedy(sd_low, sd_high, x, T, K, sample_high_seed=sample_seed, attention=attention, n_glimpses=n_glimpses)
    for key in ref:
        assert torch.equal(ref[key], orc[key]), f"oracle != reference on {key} ({name})"
    greedy = opn.two_level_greedy(sd_low, sd_high, x, T, K, attention=attention, n_glimpses=n_glimpses)
    frac = float((greedy["idx_high"] != ref["idx_high"]).float().mean())
    out = {k: v.numpy() for k, v in ref.items()}
    out.update(inputs=x.numpy(), hidden=H, n_cat=T, n_per=K, B=B, seed_low=seed, seed_high=seed + 1, sample_seed=sample_seed,
               attention=attention, n_glimpses=n_glimpses, embedding_size=E,
               margin_low=orc["margin_low"].numpy(), margin_high=orc["margin_high"].numpy())
    np.savez_compressed(os.path.join(HERE, f"pn_sample_{name}.npz"), **out)
    print(f"pn_sample_{name}: {attention} glimpses={n_glimpses} E={E} B={B} T={T} K={K} H={H}: {frac:.2f} of the sampled picks "
          f"differ from the greedy ones; min draw margin {float(orc['margin_high'].min()):.2e}")


def gen_pn_train(modelPN, name, H, T, K, B, seed, sample_seed, full=True, E=0, attention="Dot", n_glimpses=0):
    """ONE REINFORCE step of the PNHigh trainer run on the REAL modules (trainPNHigh.py:83-108; the driver class itself
    imports IPython/matplotlib and is not importable here, so its loop body is driven by hand, line for line):
    Low greedy -> latent, High sampled (multinomial routed to the stream), advantage against the first-batch critic,
    sum of log-probs, backward through the reference's own autograd graph, clip_grad_norm_, Adam(lr=0.5e-4).
    Stores every actor gradient (before clipping) and the weights after the step — in full for the small case, as norms,
    seeded samples and seeded projections for H = 256 (4 MB per set otherwise) — after checking that oracle/pn_train.py
    reproduces them."""
    from oracle import pn_train as optr
    # E != 0: embeddingTag=1 (trainPNHigh.py:197-201: embedding_size = 20) — rows [category | 8 floats], embedding1 trained too
    # attention / n_glimpses: the forms the reference's configurations switch off ('Bahdanau' attention, glimpse rounds,
    # modelPN.py:80-90,208-211) — both nets built with them, the High net trained through them
    sd_low = opn.make_state_dict(H, seed, embedding_size=E, n_cat=T, attention=attention)
    sd_high = opn.make_state_dict(H, seed + 1, embedding_size=E, n_cat=T, attention=attention)
    L = T * K

    def build(level, sd):
        m = modelPN.CombinatorialRL(E, H, L, n_glimpses, 10, 1, modelPN.reward, attention, K, T, use_cuda=False, level=level)
        m.load_state_dict(sd, strict=True)
        return m

    low, high = build("Low", sd_low), build("High", sd_high)
    x = pn_inputs(B, T, K, seed + 2)
    if E:
        cat = torch.arange(T).repeat_interleave(K).float().view(1, L, 1).expand(B, L, 1)     # loadData.py:130-148: column 0 = the category
        x = torch.cat([cat, x], 2).contiguous()
    state = {"k": 0}
    real = torch.Tensor.multinomial

    def routed(self, num_samples=1, replacement=False, generator=None):
        idx, _ = opn.multinomial_from_stream(self.detach(), state["k"], T, K, sample_seed)
        state["k"] += 1
        return idx.view(-1, 1)

    torch.Tensor.multinomial = routed
    optim = torch.optim.Adam(high.actor.parameters(), lr=0.5e-4)                     # trainPNHigh.py:62
    try:
        low.train(), high.train()                                                    # :79-80
        with contextlib.redirect_stdout(io.StringIO()):
            with torch.no_grad():
                _, _, _, _, latent = low(x, None, sample="greedy", training="SL")    # :83
            R, probs, actions, idxs, _ = high(x, None, latent)                       # :84
        critic = R.mean()                                                            # :87-88 (batch_id == 0)
        advantage = R - critic                                                       # :92
        logprobs = 0
        for prob in probs:                                                           # :94-97
            logprobs = logprobs + torch.log(prob)
        logprobs[logprobs < -1000] = 0.                                              # :98
        actor_loss = (advantage * logprobs).mean()                                   # :100-101
        optim.zero_grad()
        actor_loss.backward()                                                        # :103-104
        grads = {"actor." + n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))     # (the glimpse module without rounds)
                 for n, p in high.actor.named_parameters()}
        norm = torch.nn.utils.clip_grad_norm_([p for p in high.actor.parameters() if p.grad is not None], 2.0, norm_type=2)   # :105-106
        optim.step()                                                                 # :108
    finally:
        torch.Tensor.multinomial = real
    new_params = {"actor." + n: p.detach().clone() for n, p in high.actor.named_parameters()}
    idx_high = torch.stack(idxs, 1)
    orc = optr.train_step(sd_low, sd_high, x, T, K, sample_seed, attention=attention, n_glimpses=n_glimpses)
    assert torch.equal(orc["idx_high"], idx_high) and torch.equal(orc["R"], R.detach())
    worst = 0.0
    KEYS = optr.param_keys(sd_high)
    assert set(grads) == set(KEYS)
    for k in KEYS:
        g, og = grads[k], orc["grads"][k]
        rel = float((g - og).norm() / (g.norm() + 1e-20))
        worst = max(worst, rel)
        assert rel < 2e-4, f"oracle gradient of {k} differs from the reference's: rel {rel:.2e} ({name})"
        # Adam's first step is lr * g / (|g| + eps): for gradients near the 1e-8 eps it is ill-conditioned (a sign flip
        # of a 1e-9 gradient moves the weight by 1e-4 * ...), so weights are compared where |g| is clear of it
        clear = g.abs() > 1e-5
        assert float(((new_params[k] - orc["new_params"][k]).abs() * clear).max()) < 2e-6, k
        assert float((new_params[k] - orc["new_params"][k]).abs().max()) <= 1.01e-4, k
    assert abs(float(norm) - float(orc["grad_norm"])) < 1e-4 * max(1.0, float(norm))
    out = {"hidden": H, "n_cat": T, "n_per": K, "B": B, "seed_low": seed, "seed_high": seed + 1, "seed_inputs": seed + 2,
           "sample_seed": sample_seed, "idx_low": orc["idx_low"].numpy(), "idx_high": idx_high.numpy(), "R": R.detach().numpy(),
           "loss": float(actor_loss), "grad_norm": float(norm), "margin_low": orc["margin_low"].numpy(),
           "margin_high": orc["margin_high"].numpy(), "win_low": orc["win_low"].numpy(), "embedding_size": E, "attention": attention, "n_glimpses": n_glimpses}
    g = torch.Generator().manual_seed(seed)
    for k in KEYS:
        short = k.replace("actor.", "").replace(".", "_")
        gk, pk = grads[k].flatten(), new_params[k].flatten()
        if full:
            out["grad_" + short], out["new_" + short] = grads[k].numpy(), new_params[k].numpy()
        else:
            pos = torch.randint(0, gk.numel(), (64,), generator=g)
            vec = torch.randn(gk.numel(), generator=g)
            out["gradnorm_" + short] = float(gk.norm())
            out["gradpos_" + short], out["gradval_" + short] = pos.numpy(), gk[pos].numpy()
            out["gradproj_" + short] = float((gk.double() * vec.double()).sum())
            out["newval_" + short] = pk[pos].numpy()
            out["newproj_" + short] = float((pk.double() * vec.double()).sum())
    np.savez_compressed(os.path.join(HERE, f"pn_train_{name}.npz"), **out)
    print(f"pn_train_{name}: B={B} T={T} K={K} H={H}: loss {float(actor_loss):+.5f}, grad norm {float(norm):.4f}, "
          f"max rel. gradient error oracle vs reference {worst:.1e}")


def gen_reward(modelPN):
    """reward/calc known-answer cases (modelPN.py:15-72)."""
    rng = np.random.default_rng(7)
    T, B = 5, 6
    act = np.zeros((B, T, 8), np.float32)
    act[:, :, 0:2] = rng.random((B, T, 2), dtype=np.float32)
    act[:, :, 2:4] = 0.9 + 0.1 * rng.random((B, T, 2), dtype=np.float32)
    act[:, 0, 4:8] = [0.5, 1.0, 0.5, 1.0]
    act[1, 0, 4:8] = [0.99, 1.0, 0.5, 1.0]          # constraint 0 violated (product too small)
    act[2, 0, 4:8] = [0.99, 1.0, 0.1, 0.2]          # both violated (too small / too large)
    act[3, 2, 0:4] = [0, 1, 1, 1]                   # one dummy row
    act[4, 1:, 0:4] = [0, 1, 1, 1]                  # all but one dummy
    act[5, :, 2] = 1.0                              # product exactly 1.0 == hi bound (not violated)
    actions = [torch.from_numpy(act[:, t]) for t in range(T)]
    with contextlib.redirect_stdout(io.StringIO()):
        r_low = modelPN.reward(actions, None, T, USE_CUDA=False, level="Low", embedding_size=0)
        r_high = modelPN.reward(actions, None, T, USE_CUDA=False, level="High", embedding_size=0)
    assert torch.equal(opn.reward(actions, "Low"), r_low) and torch.equal(opn.reward(actions, "High"), r_high)
    np.savez_compressed(os.path.join(HERE, "reward.npz"), actions=act, R_low=r_low.numpy(), R_high=r_high.numpy())
    print("reward:", r_low.tolist(), r_high.tolist())


def gen_data(loadData_mod, ML2PN_mod):
    T, S, P, K = 6, 40, 16, 3
    ds = synth.make_dataset(T, S, P, seed=3, tasks_per_problem=3, lo_range=(0.80, 0.955))
    rng = np.random.default_rng(5)
    rank_shared = rng.permutation(S).tolist()
    rank_each = [rng.permutation(S).tolist() for _ in range(P)]
    # The reference spins forever (loadData.py:137-138) when a PRESENT category has no feasible
    # service, so check with the oracle first that this dataset has none such, but does exercise
    # the padding path (fewer than K feasible).
    n_pad = 0
    for nodes, rk in zip(ds["nodefeatures"] * 2, [rank_shared] * P + rank_each):
        _, present = odata.problem_constraints(nodes, T)
        _, picked = odata.reduce_candidates(rk, nodes, ds["serviceFeature"], K)
        assert all(len(picked[c - 1]) > 0 for c in present), "dataset would hang the reference"
        n_pad += sum(len(picked[c - 1]) < K for c in present)
    assert n_pad > 0, "dataset does not exercise the padding path"
    n_test = P // 4
    fx = {"T": T, "S": S, "P": P, "K": K, "dataset": ds, "rank_shared": rank_shared, "rank_each": rank_each}
    old = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            synth.write_dataset(tmp, "QWS", ds)
            os.makedirs("solutions/pretrained")
            # ---- loadData (loadData.py:14-69)
            nf, sfl, ei, eis, eas, labels, inv = loadData_mod.loadData("QWS")
            src_dst, w = odata.service_graph(ds["labels"])
            assert nf == odata.node_rows(ds["nodefeatures"]) and sfl == odata.service_rows(ds["serviceFeature"])
            assert eis == src_dst and np.allclose(eas, w, rtol=0, atol=0)
            fx["loadData"] = {"nodefeatures": nf, "serviceFeatureList": sfl, "edge_indices_service": eis,
                              "edge_attrs_service": eas}
            # ---- loadDataPN, shared ranking, shuffle patched to "sort by rank" => exact rows
            pos = {s: i for i, s in enumerate(rank_shared)}
            real_shuffle = np.random.shuffle
            np.random.shuffle = lambda lst: lst.sort(key=pos.get)
            with open("solutions/pretrained/QWS-ML.txt", "w") as f:
                json.dump([rank_shared] * P, f)
            rows_shared, lab = loadData_mod.loadDataPN(epoch=-1, dataset="QWS", serviceNumber=K)
            mine, lab2 = odata.load_data_pn(ds["nodefeatures"], ds["serviceFeature"], [rank_shared] * P,
                                            ds["minCostList"], K)
            assert rows_shared == mine and lab == lab2, "oracle loadDataPN != reference (shared ranking)"
            fx["rows_shared"] = rows_shared
            # ---- loadDataPN, per-problem rankings, real shuffle: compare the distinct rows per category
            np.random.shuffle = real_shuffle
            np.random.seed(0)
            with open("solutions/pretrained/QWS-ML.txt", "w") as f:
                json.dump(rank_each, f)
            rows_each, _ = loadData_mod.loadDataPN(epoch=-1, dataset="QWS", serviceNumber=K)
            mine_each, _ = odata.load_data_pn(ds["nodefeatures"], ds["serviceFeature"], rank_each,
                                              ds["minCostList"], K)
            for a, b in zip(rows_each, mine_each):
                for c in range(T):
                    sa = {tuple(r) for r in a[c * K:(c + 1) * K]}
                    sb = {tuple(r) for r in b[c * K:(c + 1) * K]}
                    assert sa == sb, "oracle loadDataPN candidate sets != reference"
            fx["rows_each_sets"] = [[sorted({tuple(r) for r in a[c * K:(c + 1) * K]}) for c in range(T)]
                                    for a in rows_each]
            # ---- ML2PN.check (ML2PN.py:15-56): actions = first candidate of every window of the
            # shared-ranking reduction, for the test quarter
            acts = [[rows_shared[P - n_test + j][c * K][1:] for j in range(n_test)] for c in range(T)]
            with open("solutions/pretrained/QWS-PNHigh.txt", "w") as f:
                json.dump(acts, f)
            with open("solutions/pretrained/QWS-ML.txt", "w") as f:
                json.dump([rank_shared] * P, f)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                ML2PN_mod.check("QWS", T, -1)
            printed = buf.getvalue().strip().split()
            score = float(printed[-1])
            k1, _ = odata.load_data_pn(ds["nodefeatures"], ds["serviceFeature"], [rank_shared] * P,
                                       ds["minCostList"], 1)
            mine_score = odata.check(k1, ds["minCostList"], acts, T)
            assert printed[0] == "-1" and abs(mine_score - score) < 1e-12, (mine_score, score)
            fx["check"] = {"actions": acts, "score": score, "printed": buf.getvalue()}
        finally:
            np.random.shuffle = real_shuffle
            os.chdir(old)
    with open(os.path.join(HERE, "data_small.json"), "w") as f:
        json.dump(fx, f)
    print("data_small: check score", fx["check"]["score"])


def gen_ml(modelML, name, hidden, emb, n_gin, n_gcn, T, S, B, seed, n_t=3, degree=6, is_services=True):
    table = synth.make_service_table(T, S, seed, degree=degree)
    batch = synth.make_problem_batch(table, B, seed + 1, tasks_per_problem=n_t)
    sd = oml.make_state_dict(hidden, emb, n_gin, n_gcn, seed + 2)
    net = modelML.Net(hidden, S, emb, n_gin, n_gcn, isServices=is_services, dropout=0.0)
    net.load_state_dict(sd, strict=True)        # pins key names + shapes
    net.eval()
    # what a PyG batch of B graphs holds (trainML.py:109-114): B copies of the service graph
    xs = torch.from_numpy(table.x_service)
    eis = torch.from_numpy(table.edge_index)
    eas = torch.from_numpy(table.edge_attr)
    data = oml.make_data(torch.from_numpy(batch.x), torch.from_numpy(batch.edge_index),
                         torch.from_numpy(batch.batch),
                         xs.repeat(B, 1), torch.cat([eis + b * S for b in range(B)], 1), eas.repeat(B))
    with torch.no_grad():
        ref = net(data)
    one = oml.make_data(data.x, data.edge_index, data.batch, xs, eis, eas)
    orc = oml.net_forward(sd, one, n_gin, n_gcn, is_services)
    err = float((ref - orc).abs().max())
    assert err <= 1e-6, f"oracle Net != reference glue: {err}"
    np.savez_compressed(
        os.path.join(HERE, f"ml_{name}.npz"), scores=ref.numpy(), ranking=oml.rank_services(ref).numpy(),
        is_services=is_services, hidden=hidden, emb=emb, n_gin=n_gin, n_gcn=n_gcn, T=T, S=S, B=B, seed=seed, n_t=n_t, degree=degree,
        x=batch.x, edge_index=batch.edge_index, batch=batch.batch, x_service=table.x_service,
        edge_index_service=table.edge_index, edge_attr_service=table.edge_attr)
    print(f"ml_{name}: B={B} S={S} max|ref-oracle|={err:.2e}")


def gen_ml_pygbatch(modelML, name, hidden, emb, n_gin, n_gcn, T, S, seed, n_t=3, degree=6):
    """The reference's forward on a batch of TWO graphs (DataLoader(batch_size=2), trainML.py:121-122) assembled the
    way torch_geometric 1.7.0 would assemble it: x_service concatenated (2 copies), edge_index_service of copy 1
    offset by graph 0's WORKFLOW node count (Data.__inc__ -> num_nodes = x.size(0); from memory of the 1.7.0 source —
    see oracle/ml.py).  The real Net glue runs on it (stand-in convs); the oracle's literal batched form must agree."""
    B = 2
    table = synth.make_service_table(T, S, seed, degree=degree)
    batch = synth.make_problem_batch(table, B, seed + 1, tasks_per_problem=n_t)
    sd = oml.make_state_dict(hidden, emb, n_gin, n_gcn, seed + 2)
    net = modelML.Net(hidden, S, emb, n_gin, n_gcn, isServices=True, dropout=0.0)
    net.load_state_dict(sd, strict=True)
    net.eval()
    xs, eis, eas = (torch.from_numpy(a) for a in (table.x_service, table.edge_index, table.edge_attr))
    nodes = np.bincount(batch.batch, minlength=B)
    offsets = np.concatenate([[0], np.cumsum(nodes)[:-1]])
    ei_b, ea_b = oml.pyg_batch_service_edges(eis, eas, offsets)
    data = oml.make_data(torch.from_numpy(batch.x), torch.from_numpy(batch.edge_index), torch.from_numpy(batch.batch),
                         xs.repeat(B, 1), ei_b, ea_b)
    with torch.no_grad():
        ref = net(data)
    orc = oml.net_forward_batched(sd, data, n_gin, n_gcn, S)
    err = float((ref - orc).abs().max())
    assert err <= 1e-6, f"oracle batched Net != reference glue: {err}"
    clean = oml.net_forward(sd, oml.make_data(data.x, data.edge_index, data.batch, xs, eis, eas), n_gin, n_gcn)
    np.savez_compressed(
        os.path.join(HERE, f"ml_{name}.npz"), scores=ref.numpy(), hidden=hidden, emb=emb, n_gin=n_gin, n_gcn=n_gcn, T=T, S=S,
        B=B, seed=seed, x=batch.x, edge_index=batch.edge_index, batch=batch.batch, x_service=table.x_service,
        edge_index_service=table.edge_index, edge_attr_service=table.edge_attr, offsets=offsets,
        scores_single_copy=clean.numpy())
    print(f"ml_{name}: pyg-1.7.0-style batch of 2, max|ref-oracle|={err:.2e}, max|batched - single-copy scores|="
          f"{float((ref - clean).abs().max()):.3f}")


def gen_ml_train(modelML, name, hidden, emb, n_gin, n_gcn, T, S, seed, lr, n_t=3, degree=6, steps=2, is_services=True):
    """``steps`` consecutive batches of TrainML.train (trainML.py:34-47) on the REAL Net glue (stand-in
    convs) under autograd — model.train(), BCELoss, backward, Adam(lr) — each on a torch_geometric-1.7.0-style batch of two
    graphs.  Stores, for the first and the last step, the loss, every gradient, the weights after the step and the
    BatchNorm running statistics; the autograd oracle (oracle/ml_train.py) must reproduce them."""
    from oracle import ml_train as omt
    B = 2
    table = synth.make_service_table(T, S, seed, degree=degree)
    sd = oml.make_state_dict(hidden, emb, n_gin, n_gcn, seed + 2)
    net = modelML.Net(hidden, S, emb, n_gin, n_gcn, isServices=is_services, dropout=0.0)   # False: the ablation of modelML.py:157-162
    net.load_state_dict(sd, strict=True)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=lr)                                        # trainML.py:130
    crit = torch.nn.BCELoss()                                                              # :28
    xs, eis, eas = (torch.from_numpy(a) for a in (table.x_service, table.edge_index, table.edge_attr))
    keys = omt.trainable_keys(sd, n_gin, n_gcn, is_services)
    cur, st, out = dict(sd), None, {}
    g = torch.Generator().manual_seed(seed + 7)
    for step in range(1, steps + 1):
        batch = synth.make_problem_batch(table, B, seed + 10 * step, tasks_per_problem=n_t)
        nodes = np.bincount(batch.batch, minlength=B)
        offsets = np.concatenate([[0], np.cumsum(nodes)[:-1]])
        ei_b, ea_b = oml.pyg_batch_service_edges(eis, eas, offsets)
        data = oml.make_data(torch.from_numpy(batch.x), torch.from_numpy(batch.edge_index), torch.from_numpy(batch.batch),
                             xs.repeat(B, 1), ei_b, ea_b)
        y = (torch.rand(B * S, generator=g) < 0.02).float()                                # labels: a few services per problem
        opt.zero_grad()                                                                    # :40
        x = net(data).squeeze()                                                            # :41
        loss = crit(x, y.view(x.size(0), x.size(1)))                                       # :42
        loss.backward()                                                                    # :43
        grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        assert set(grads) == set(keys), sorted(set(grads) ^ set(keys))
        opt.step()                                                                         # :45
        ref_sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
        orc = omt.train_step(cur, data, y, n_gin, n_gcn, S, lr, st, step, is_services)
        assert abs(float(orc["loss"]) - float(loss.detach())) <= 1e-6 * max(1.0, abs(float(loss.detach())))
        gmax = max(float(grads[k].abs().max()) for k in keys)
        for k in keys:
            # a bias in front of a training-mode BatchNorm has an exactly zero gradient: rounding noise on both sides,
            # so the bound is relative to the parameter's own scale OR 1e-6 of the largest gradient of the step
            scale = float(grads[k].abs().max())
            assert float((orc["grads"][k] - grads[k]).abs().max()) <= 2e-5 * scale + 1e-6 * gmax, (k, step)
            clear = grads[k].abs() > 1e-6       # Adam's step is sign-like near g = 0 (eps = 1e-8): ill-conditioned there
            if bool(clear.any()):
                assert float((orc["new_params"][k] - ref_sd[k])[clear].abs().max()) <= 2e-6, (k, step)
        for pre, (rm, rv) in orc["running"].items():
            assert float((rm - ref_sd[pre + ".running_mean"]).abs().max()) <= 1e-5, pre
            assert float((rv - ref_sd[pre + ".running_var"]).abs().max()) <= 1e-5 * max(1.0, float(rv.abs().max())), pre
        for tag in (["first"] if step == 1 else []) + (["last"] if step == steps and steps > 1 else []):
            out.update({f"{tag}_x": batch.x, f"{tag}_edge_index": batch.edge_index, f"{tag}_batch": batch.batch,
                        f"{tag}_offsets": offsets, f"{tag}_y": y.numpy(), f"{tag}_loss": float(loss.detach()), f"{tag}_scores": x.detach().numpy()})
            for k in keys:
                out[f"{tag}_grad/{k}"] = grads[k].numpy()
                out[f"{tag}_param/{k}"] = ref_sd[k].numpy()
            for pre in orc["running"]:
                out[f"{tag}_running_mean/{pre}"] = ref_sd[pre + ".running_mean"].numpy()
                out[f"{tag}_running_var/{pre}"] = ref_sd[pre + ".running_var"].numpy()
        if step == steps - 1:
            # the state the LAST step starts from (so that a test can replay that step alone)
            for k, v in ref_sd.items():
                if v.dtype.is_floating_point:
                    out[f"before_last/{k}"] = v.numpy()
            for k in keys:
                m, v = opt.state[dict(net.named_parameters())[k]]["exp_avg"], opt.state[dict(net.named_parameters())[k]]["exp_avg_sq"]
                out[f"before_last_m/{k}"], out[f"before_last_v/{k}"] = m.clone().numpy(), v.clone().numpy()
        # carry on from the REFERENCE's state (not the oracle's) so that errors cannot accumulate unnoticed
        cur = {k: v.clone() for k, v in ref_sd.items()}
        named = dict(net.named_parameters())
        st = {k: (opt.state[named[k]]["exp_avg"].clone(), opt.state[named[k]]["exp_avg_sq"].clone()) for k in keys}
    np.savez_compressed(os.path.join(HERE, f"ml_train_{name}.npz"), hidden=hidden, emb=emb, n_gin=n_gin, n_gcn=n_gcn, T=T, S=S,
                        B=B, seed=seed, lr=lr, steps=steps, is_services=int(is_services), x_service=table.x_service, edge_index_service=table.edge_index,
                        edge_attr_service=table.edge_attr, **out)
    print(f"ml_train_{name}: {steps} steps, loss first/last = {out['first_loss']:.6f} / {out.get('last_loss', out['first_loss']):.6f}")


def gen_hand_graph():
    """G6: a 5-node hand-checkable graph for the aggregate ops (expected values by oracle, which
    the stand-in cross-checks)."""
    x = torch.arange(10, dtype=torch.float32).view(5, 2) + 1
    ei = torch.tensor([[0, 1, 1, 2, 3, 4, 4, 2], [1, 0, 2, 1, 4, 3, 4, 2]])
    w = torch.tensor([0.5, 0.25, 1.0, 2.0, 0.75, 0.125, 3.0, 0.0])
    weight = torch.tensor([[1.0, -1.0, 0.5], [0.25, 2.0, -0.5]])
    bias = torch.tensor([0.1, -0.2, 0.3])
    conv = pyg_standin.GCNConv(2, 3)
    with torch.no_grad():
        conv.weight.copy_(weight)
        conv.bias.copy_(bias)
        ref = conv(x, ei, w)
    orc = oml.gcn_conv(x, ei, w, weight, bias)
    assert torch.equal(ref, orc)
    gin_sum = oml.scatter_sum(x[ei[0]], ei[1], 5) + (1 + 0.1) * x
    np.savez_compressed(os.path.join(HERE, "hand_graph.npz"), x=x.numpy(), edge_index=ei.numpy(), w=w.numpy(),
                        weight=weight.numpy(), bias=bias.numpy(), gcn=ref.numpy(), gin_eps=0.1,
                        gin_pre=gin_sum.numpy())
    print("hand_graph ok")


def main():
    import signal
    signal.alarm(600)                           # the reference has an infinite-loop path; never hang
    modelPN, modelML, loadData_mod, ML2PN_mod = import_reference()
    torch.set_num_threads(1)                    # fixtures must not depend on the thread count
    gen_pn(modelPN, "small", H=32, T=6, K=3, B=4, seed=11)
    gen_pn(modelPN, "dummy", H=32, T=7, K=2, B=3, seed=21, dummy_every=3)
    gen_pn(modelPN, "qws", H=256, T=47, K=5, B=8, seed=31)
    gen_pn(modelPN, "normal", H=256, T=50, K=10, B=4, seed=41)
    gen_pn(modelPN, "saturated", H=256, T=12, K=5, B=8, seed=51, weight_scale=6.0)
    gen_pn_sample(modelPN, "small", H=32, T=6, K=3, B=8, seed=91, sample_seed=12345)
    gen_pn_sample(modelPN, "qws", H=256, T=47, K=5, B=64, seed=95, sample_seed=987654321)
    gen_pn_sample_forms(modelPN, "bahdanau_g1_small", H=32, T=6, K=3, B=8, seed=151, sample_seed=2024, attention="Bahdanau", n_glimpses=1)
    gen_pn_sample_forms(modelPN, "dot_g2_small", H=32, T=6, K=3, B=8, seed=153, sample_seed=2025, attention="Dot", n_glimpses=2)
    gen_pn_sample_forms(modelPN, "embed_small", H=32, T=6, K=3, B=8, seed=155, sample_seed=2026, E=4)
    gen_pn_sample_forms(modelPN, "bahdanau_g1_qws", H=256, T=47, K=5, B=16, seed=157, sample_seed=2027, attention="Bahdanau", n_glimpses=1)
    gen_pn_sample_forms(modelPN, "embed_qws", H=256, T=47, K=5, B=16, seed=159, sample_seed=2028, E=20)
    gen_pn_train(modelPN, "small", H=32, T=6, K=3, B=8, seed=101, sample_seed=4242, full=True)
    gen_pn_train(modelPN, "qws", H=256, T=47, K=5, B=32, seed=105, sample_seed=777, full=False)
    gen_pn_train(modelPN, "embed_small", H=32, T=6, K=3, B=8, seed=161, sample_seed=4343, full=True, E=4)
    gen_pn_train(modelPN, "embed_qws", H=256, T=47, K=5, B=16, seed=165, sample_seed=778, full=False, E=20)
    gen_pn_train(modelPN, "bahdanau_g1_small", H=32, T=6, K=3, B=8, seed=171, sample_seed=4444, full=True, attention="Bahdanau", n_glimpses=1)
    gen_pn_train(modelPN, "dot_g2_small", H=32, T=6, K=3, B=8, seed=173, sample_seed=4445, full=True, attention="Dot", n_glimpses=2)
    gen_pn_train(modelPN, "bahdanau_g0_small", H=32, T=6, K=3, B=8, seed=175, sample_seed=4446, full=True, attention="Bahdanau", n_glimpses=0)
    gen_pn_train(modelPN, "bahdanau_g1_qws", H=256, T=47, K=5, B=8, seed=177, sample_seed=779, full=False, attention="Bahdanau", n_glimpses=1)
    gen_pn_attn(modelPN, "dot_g1_small", H=32, T=6, K=3, B=6, seed=111, attention="Dot", n_glimpses=1)
    gen_pn_attn(modelPN, "bahdanau_g0_small", H=32, T=6, K=3, B=6, seed=113, attention="Bahdanau", n_glimpses=0)
    gen_pn_attn(modelPN, "bahdanau_g2_small", H=32, T=6, K=3, B=6, seed=115, attention="Bahdanau", n_glimpses=2)
    gen_pn_attn(modelPN, "dot_g1_qws", H=256, T=47, K=5, B=16, seed=117, attention="Dot", n_glimpses=1)
    gen_pn_attn(modelPN, "bahdanau_g1_qws", H=256, T=47, K=5, B=16, seed=119, attention="Bahdanau", n_glimpses=1)
    gen_pn_embed(modelPN, "small", H=32, T=6, K=3, B=6, seed=141, E=4)
    gen_pn_embed(modelPN, "qws", H=256, T=47, K=5, B=16, seed=145, E=20)
    gen_reward(modelPN)
    gen_data(loadData_mod, ML2PN_mod)
    gen_ml(modelML, "tiny", hidden=16, emb=8, n_gin=2, n_gcn=2, T=6, S=40, B=2, seed=51)
    gen_ml(modelML, "qws", hidden=128, emb=20, n_gin=2, n_gcn=2, T=47, S=300, B=2, seed=61, n_t=10, degree=8)
    gen_ml(modelML, "normal", hidden=128, emb=20, n_gin=2, n_gcn=4, T=50, S=250, B=1, seed=71, n_t=10, degree=8)
    gen_ml(modelML, "noservices", hidden=128, emb=20, n_gin=2, n_gcn=2, T=47, S=300, B=3, seed=81, n_t=10, degree=8,
           is_services=False)                                                     # modelML.py:157-162
    gen_ml_pygbatch(modelML, "pygbatch", hidden=128, emb=20, n_gin=2, n_gcn=2, T=47, S=300, seed=81, n_t=10, degree=8)
    gen_ml_train(modelML, "tiny", hidden=16, emb=8, n_gin=2, n_gcn=2, T=6, S=40, seed=121, lr=1e-3, steps=3)
    gen_ml_train(modelML, "qws", hidden=128, emb=20, n_gin=2, n_gcn=2, T=47, S=300, seed=131, lr=1e-3, n_t=10, degree=8, steps=1)
    gen_ml_train(modelML, "noservices", hidden=16, emb=8, n_gin=2, n_gcn=2, T=6, S=40, seed=141, lr=1e-3, steps=2, is_services=False)
    gen_hand_graph()
    import manifest
    manifest.update()                           # sha256 + key lists of every fixture: what the CPU suite holds the committed files to


if __name__ == "__main__":
    main()
