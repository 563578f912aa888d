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


ATTN_KEYS = tuple(f"actor.{which}.{name}" for which in ("pointer", "glimpse")
                  for name in ("W_query.weight", "W_query.bias", "W_ref.weight", "W_ref.bias", "V"))


def param_keys(sd):
    """PARAM_KEYS, plus the category embedding table when the net has one (embedding_size != 0, modelPN.py:153-154) and the
    parameters of the two Attention modules when they have any ('Bahdanau', modelPN.py:82-90; the glimpse module's stay
    without gradient when n_glimpses = 0: autograd returns zeros for them here, the reference's .grad is None)."""
    return (PARAM_KEYS + (("actor.embedding1.weight",) if "actor.embedding1.weight" in sd else ())
            + tuple(k for k in ATTN_KEYS if k in sd))


def _cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    gates = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, g, o = gates.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def _attention(p, which, name, query, enc_out, use_tanh, C):
    """Attention.forward (modelPN.py:92-122) on the parameter dict: -> (ref' [B,L,H], logits [B,L])."""
    if name == "Bahdanau":                                      # the reference's own operations in its order (the summation order of the
        refp = enc_out.permute(0, 2, 1)                         # products decides the last bits of a 47-step gradient)            :104
        q = F.linear(query, p[f"actor.{which}.W_query.weight"], p[f"actor.{which}.W_query.bias"]).unsqueeze(2)       # :105
        refp = F.conv1d(refp, p[f"actor.{which}.W_ref.weight"], p[f"actor.{which}.W_ref.bias"])                     # :106
        V = p[f"actor.{which}.V"].unsqueeze(0).unsqueeze(0).repeat(enc_out.shape[0], 1, 1)                          # :108
        logits = torch.bmm(V, torch.tanh(q.repeat(1, 1, refp.shape[2]) + refp)).squeeze(1)                          # :107-109
        refp = refp.permute(0, 2, 1)                            # [B,L,H] for the caller
    elif name == "Dot":
        logits = torch.bmm(enc_out, query.unsqueeze(2)).squeeze(2)                                                 # :112-113
        refp = enc_out
    else:
        raise NotImplementedError(name)
    if use_tanh:
        logits = C * torch.tanh(logits)                                                                            # :119-120
    return refp, logits


def pick_log_probs(params, inputs, idx, n_cat, n_per, latent_win=None, C=10.0, use_tanh=True, attention="Dot", n_glimpses=0):
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
        if attention == "Dot" and n_glimpses == 0:
            win = enc_out[:, k * n_per:(k + 1) * n_per]                                              # only the window survives the mask (:220-222)
            z = torch.bmm(win, h.unsqueeze(2)).squeeze(2)                                            # :112-113
            if use_tanh:
                z = C * torch.tanh(z)                                                                # :119-120
        else:                                                                                        # :207-214, the picks of the earlier steps masked
            chosen = torch.zeros(B, L, dtype=torch.bool)
            if k:
                chosen[rows.unsqueeze(1), idx[:, :k]] = True
            query = h
            for _ in range(n_glimpses):
                refp, gl = _attention(p, "glimpse", attention, query, enc_out, False, C)
                gl = gl.masked_fill(chosen, float("-inf"))
                query = torch.bmm(F.softmax(gl, dim=1).unsqueeze(1), refp).squeeze(1)                # :209
            _, z = _attention(p, "pointer", attention, query, enc_out, use_tanh, C)
            z = z[:, k * n_per:(k + 1) * n_per]                                                      # (no earlier pick lies in this window)
        if latent_win is not None:
            z = z + latent_win[:, k]                                                                 # :215-216
        lp = F.log_softmax(z, dim=1)                                                                 # :224
        r = idx[:, k] - k * n_per
        logp.append(lp[rows, r])
        prob.append(torch.exp(lp[rows, r]))
        x = embedded[rows, idx[:, k]]                                                                # :235
    return torch.stack(logp, 1), torch.stack(prob, 1)


def reinforce_step(sd_high, inputs, idx, R, n_cat, n_per, latent_win, critic=None, beta=0.9, lr=0.5e-4, max_grad_norm=2.0,
                   adam_state=None, step=1, attention="Dot", n_glimpses=0):
    """trainPNHigh.py:87-108 for ONE batch.  Returns dict(loss, critic, advantage, grads {name: tensor} BEFORE clipping,
    grad_norm, new_params {name: tensor} after clip + Adam, adam_state)."""
    KEYS = param_keys(sd_high)
    params = {k: sd_high[k].clone().requires_grad_(True) for k in KEYS}
    logp, prob = pick_log_probs(params, inputs, idx, n_cat, n_per, latent_win, attention=attention, n_glimpses=n_glimpses)
    critic = R.mean() if critic is None else critic * beta + (1.0 - beta) * R.mean()                # :87-90
    advantage = R - critic                                                                           # :92
    # :94-98 — the reference sums log(prob) of the returned probabilities; log_softmax at the pick is the same number
    logprobs = logp.sum(1)
    logprobs = torch.where(logprobs < -1000, torch.zeros_like(logprobs), logprobs)
    loss = (advantage.detach() * logprobs).mean()                                                    # :100-101
    grads = torch.autograd.grad(loss, [params[k] for k in KEYS], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(params[k])) for k, g in zip(KEYS, grads)}   # (the glimpse module without rounds)
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


def train_step(sd_low, sd_high, inputs, n_cat, n_per, sample_seed, attention="Dot", n_glimpses=0, **kw):
    """The whole step of trainPNHigh.py:83-108: Low greedy -> latent, High sampled (stream of ``sample_seed``), REINFORCE."""
    fwd = opn.two_level_greedy(sd_low, sd_high, inputs, n_cat, n_per, sample_high_seed=sample_seed, attention=attention,
                               n_glimpses=n_glimpses)
    out = reinforce_step(sd_high, inputs, fwd["idx_high"], fwd["R"], n_cat, n_per, fwd["win_low"], attention=attention,
                         n_glimpses=n_glimpses, **kw)
    out.update(idx_low=fwd["idx_low"], idx_high=fwd["idx_high"], R=fwd["R"], win_low=fwd["win_low"],
               margin_low=fwd["margin_low"], margin_high=fwd["margin_high"])
    return out
