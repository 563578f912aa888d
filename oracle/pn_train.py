"""Oracle (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py): ONE REINFORCE training step of the High-level pointer
network, restated on PyTorch-CPU autograd from /root/reference/src/models/trainPNHigh.py:76-112.

    Low greedy ("SL") -> latent                                   trainPNHigh.py:83
    R, probs, actions, idxs = High(inputs, labs, latent)          :84   (sample="sample": draws from the stream, oracle/pn.py)
    critic = R.mean() on the first batch, else critic*beta + (1-beta)*R.mean()      :87-90
    advantage = R - critic ; logprobs = sum_k log(prob_k) ; logprobs[logprobs < -1000] = 0      :92-98
    actor_loss = (advantage * logprobs).mean() ; backward ; clip_grad_norm_(actor, max_grad_norm) ; Adam step      :100-108

The forward that autograd differentiates is the reference's arithmetic (modelPN.py:175-241) with the picks GIVEN (they
were drawn by the sampled forward and are constants of the graph): embedding2 -> encoder LSTM -> T x {decoder LSTM cell,
dot attention over all L, C*tanh, + latent, window mask, softmax, gather of the pick's probability, next input}.
Pinned against the real reference by tests/golden/make_golden.py::gen_pn_train (gradients of every actor parameter and
the weights after the Adam step).
"""
import torch
import torch.nn.functional as F

from . import pn as opn

PARAM_KEYS = ("actor.decoder_start_input", "actor.embedding2.weight", "actor.embedding2.bias",
              "actor.encoder.weight_ih_l0", "actor.encoder.weight_hh_l0", "actor.encoder.bias_ih_l0", "actor.encoder.bias_hh_l0",
              "actor.decoder.weight_ih_l0", "actor.decoder.weight_hh_l0", "actor.decoder.bias_ih_l0", "actor.decoder.bias_hh_l0")


def param_keys(sd):
    """PARAM_KEYS, plus the category embedding table when the net has one (embedding_size != 0, modelPN.py:153-154)."""
    return PARAM_KEYS + (("actor.embedding1.weight",) if "actor.embedding1.weight" in sd else ())


def _cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    gates = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, g, o = gates.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def pick_log_probs(params, inputs, idx, n_cat, n_per, latent_win=None, C=10.0, use_tanh=True):
    """Differentiable forward with the picks given: -> (log_probs [B,T] of the picks, probs_of_picks [B,T]).
    params: dict name -> tensor (requires_grad as the caller set it); idx [B,T] int64 global positions;
    latent_win [B,T,K] (the Low net's window logits, constants) or None."""
    B, L, _ = inputs.shape
    p = params
    if "actor.embedding1.weight" in p:                                                               # :183-188 (embedding_size != 0)
        x1 = F.embedding(inputs[:, :, 0].long(), p["actor.embedding1.weight"])
        inputs = torch.cat([x1, inputs[:, :, 1:]], 2)
    embedded = F.linear(inputs, p["actor.embedding2.weight"], p["actor.embedding2.bias"])           # modelPN.py:190
    h = torch.zeros(B, embedded.shape[2])
    c = torch.zeros_like(h)
    enc = []
    for t in range(L):                                                                               # :191 (nn.LSTM, written out)
        h, c = _cell(embedded[:, t], h, c, p["actor.encoder.weight_ih_l0"], p["actor.encoder.weight_hh_l0"],
                     p["actor.encoder.bias_ih_l0"], p["actor.encoder.bias_hh_l0"])
        enc.append(h)
    enc_out = torch.stack(enc, 1)                                                                    # [B,L,H]
    x = p["actor.decoder_start_input"].unsqueeze(0).repeat(B, 1)                                     # :202
    rows = torch.arange(B)
    logp, prob = [], []
    for k in range(n_cat):                                                                           # :204
        h, c = _cell(x, h, c, p["actor.decoder.weight_ih_l0"], p["actor.decoder.weight_hh_l0"],
                     p["actor.decoder.bias_ih_l0"], p["actor.decoder.bias_hh_l0"])                   # :205
        win = enc_out[:, k * n_per:(k + 1) * n_per]                                                  # only the window survives the mask (:220-222)
        z = torch.bmm(win, h.unsqueeze(2)).squeeze(2)                                                # :112-113
        if use_tanh:
            z = C * torch.tanh(z)                                                                    # :119-120
        if latent_win is not None:
            z = z + latent_win[:, k]                                                                 # :215-216
        lp = F.log_softmax(z, dim=1)                                                                 # :224
        r = idx[:, k] - k * n_per
        logp.append(lp[rows, r])
        prob.append(torch.exp(lp[rows, r]))
        x = embedded[rows, idx[:, k]]                                                                # :235
    return torch.stack(logp, 1), torch.stack(prob, 1)


def reinforce_step(sd_high, inputs, idx, R, n_cat, n_per, latent_win, critic=None, beta=0.9, lr=0.5e-4, max_grad_norm=2.0,
                   adam_state=None, step=1):
    """trainPNHigh.py:87-108 for ONE batch.  Returns dict(loss, critic, advantage, grads {name: tensor} BEFORE clipping,
    grad_norm, new_params {name: tensor} after clip + Adam, adam_state)."""
    KEYS = param_keys(sd_high)
    params = {k: sd_high[k].clone().requires_grad_(True) for k in KEYS}
    logp, prob = pick_log_probs(params, inputs, idx, n_cat, n_per, latent_win)
    critic = R.mean() if critic is None else critic * beta + (1.0 - beta) * R.mean()                # :87-90
    advantage = R - critic                                                                           # :92
    # :94-98 — the reference sums log(prob) of the returned probabilities; log_softmax at the pick is the same number
    logprobs = logp.sum(1)
    logprobs = torch.where(logprobs < -1000, torch.zeros_like(logprobs), logprobs)
    loss = (advantage.detach() * logprobs).mean()                                                    # :100-101
    grads = torch.autograd.grad(loss, [params[k] for k in KEYS])
    grads = dict(zip(KEYS, grads))
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()                 # clip_grad_norm_(.., 2) :104-105
    coef = torch.clamp(max_grad_norm / (total + 1e-6), max=1.0)
    st = adam_state or {k: (torch.zeros_like(sd_high[k]), torch.zeros_like(sd_high[k])) for k in KEYS}
    new_params, new_state = {}, {}
    b1, b2, eps = 0.9, 0.999, 1e-8                                                                   # optim.Adam defaults (:62)
    for k in KEYS:
        g = grads[k] * coef
        m = st[k][0] * b1 + (1 - b1) * g
        v = st[k][1] * b2 + (1 - b2) * g * g
        mhat, vhat = m / (1 - b1 ** step), v / (1 - b2 ** step)
        new_params[k] = sd_high[k] - lr * mhat / (vhat.sqrt() + eps)
        new_state[k] = (m, v)
    return {"loss": loss.detach(), "critic": critic.detach(), "advantage": advantage.detach(), "grads": grads,
            "grad_norm": total, "new_params": new_params, "adam_state": new_state, "pick_prob": prob.detach()}


def train_step(sd_low, sd_high, inputs, n_cat, n_per, sample_seed, **kw):
    """The whole step of trainPNHigh.py:83-108: Low greedy -> latent, High sampled (stream of ``sample_seed``), REINFORCE."""
    fwd = opn.two_level_greedy(sd_low, sd_high, inputs, n_cat, n_per, sample_high_seed=sample_seed)
    out = reinforce_step(sd_high, inputs, fwd["idx_high"], fwd["R"], n_cat, n_per, fwd["win_low"], **kw)
    out.update(idx_low=fwd["idx_low"], idx_high=fwd["idx_high"], R=fwd["R"], win_low=fwd["win_low"],
               margin_low=fwd["margin_low"], margin_high=fwd["margin_high"])
    return out
