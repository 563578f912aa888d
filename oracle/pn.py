"""Oracle (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py): two-level pointer-network
greedy inference, restated on PyTorch-CPU fp32 from /root/reference/src/models/modelPN.py.

Pinned against the imported reference by tests/golden/make_golden.py (fixtures
tests/golden/pn_*.npz): identical idxs, identical logits bit-for-bit when both run on
this image's torch CPU build.

Everything here works on a plain ``state_dict`` (name -> tensor) with the reference's
key names (modelPN.py:154-163, prefixed ``actor.`` by CombinatorialRL, modelPN.py:267):
    actor.decoder_start_input [H]
    actor.embedding2.{weight [H,8], bias [H]}
    actor.encoder.{weight_ih_l0 [4H,H], weight_hh_l0 [4H,H], bias_ih_l0 [4H], bias_hh_l0 [4H]}
    actor.decoder.{...same...}
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

QOS_NUM = 4    # modelPN.py:11
CONS_NUM = 2   # modelPN.py:12
NEG_INF = float("-inf")


def make_state_dict(hidden, seed, in_features=8, attention="Dot", embedding_size=0, n_cat=0):
    """Deterministic weights in the reference's state_dict layout (PyTorch default-init
    ranges: U(-1/sqrt(H), 1/sqrt(H)) for LSTM and decoder_start_input, modelPN.py:163;
    U(-1/sqrt(in), 1/sqrt(in)) for the Linear).  Build-owned generator so that fixtures
    can be regenerated from a seed instead of storing 4 MB of weights."""
    g = torch.Generator().manual_seed(seed)

    def u(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    kh = 1.0 / math.sqrt(hidden)
    in_features = in_features + embedding_size          # embedding2 = Linear(embedding_size + 8, H) (modelPN.py:155)
    ke = 1.0 / math.sqrt(in_features)
    sd = {
        "actor.decoder_start_input": u((hidden,), kh),
        "actor.embedding2.weight": u((hidden, in_features), ke),
        "actor.embedding2.bias": u((hidden,), ke),
    }
    for name in ("encoder", "decoder"):
        sd[f"actor.{name}.weight_ih_l0"] = u((4 * hidden, hidden), kh)
        sd[f"actor.{name}.weight_hh_l0"] = u((4 * hidden, hidden), kh)
        sd[f"actor.{name}.bias_ih_l0"] = u((4 * hidden,), kh)
        sd[f"actor.{name}.bias_hh_l0"] = u((4 * hidden,), kh)
    if embedding_size:                     # embedding1 = nn.Embedding(sCategory, embedding_size) (modelPN.py:153-154); drawn after the rest
        sd["actor.embedding1.weight"] = torch.randn((n_cat, embedding_size), generator=g)
    if attention == "Bahdanau":            # Attention.__init__ (modelPN.py:82-90); drawn after everything else
        for name in ("pointer", "glimpse"):
            sd[f"actor.{name}.W_query.weight"] = u((hidden, hidden), kh)
            sd[f"actor.{name}.W_query.bias"] = u((hidden,), kh)
            sd[f"actor.{name}.W_ref.weight"] = u((hidden, hidden, 1), kh)
            sd[f"actor.{name}.W_ref.bias"] = u((hidden,), kh)
            sd[f"actor.{name}.V"] = u((hidden,), kh)
    return sd


def _lstm_module(sd, which, hidden):
    """nn.LSTM(hidden, hidden, batch_first=True) (modelPN.py:157-158) loaded from ``sd``."""
    m = torch.nn.LSTM(hidden, hidden, batch_first=True)
    with torch.no_grad():
        for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
            getattr(m, k).copy_(sd[f"actor.{which}.{k}"])
    m.eval()
    return m


def lstm_cell_explicit(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """One LSTM cell step written out (gate order i,f,g,o as in torch.nn.LSTM).  Used by
    tests to document the arithmetic the HIP kernels implement; the oracle proper uses
    nn.LSTM so that it is bit-identical to the reference on CPU."""
    gates = F.linear(h, w_hh, b_hh) + F.linear(x, w_ih, b_ih)
    i, f, g, o = gates.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


_M64 = (1 << 64) - 1


def stream_uniform24(seed, ctr):
    """Draw ``ctr`` of the counter-based stream of ``seed`` (splitmix64 finaliser, as oracle/woa.py's stream) as a 24-bit
    uniform in [0,1): what the sampling mode of the HIP decoder uses (include/gnnpn_hip.h, gnnpn_decode_net_t.sample)."""
    z = (int(seed) + 0x9E3779B97F4A7C15 * (int(ctr) + 1)) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    z ^= z >> 31
    return np.float32((z >> 40) * (1.0 / (1 << 24)))


def multinomial_from_stream(probs, k, n_cat, n_per, seed):
    """What ``probs.multinomial(num_samples=1)`` (modelPN.py:228) is routed to, so that a sampled forward is a pure
    function of (weights, inputs, seed): probs [B,L] is zero outside step k's window; per problem b the window's running
    sum cdf_r = p_0 + ... + p_r is formed in fp32 in candidate order, u = draw (b*T + k) of the stream, and the pick is
    the first r with u < cdf_r (the last r with p_r > 0 should rounding leave none).
    Returns (idx [B] int64 global positions, margin [B] = distance of u to the nearest cdf boundary that could change
    the pick — a draw with a tiny margin is fragile the way a near-tie is for the argmax)."""
    B = probs.shape[0]
    win = probs[:, k * n_per:(k + 1) * n_per].numpy().astype(np.float32)
    idx = np.zeros(B, np.int64)
    margin = np.zeros(B, np.float32)
    for b in range(B):
        u = stream_uniform24(seed, b * n_cat + k)
        cdf, pick, last_pos, bounds = np.float32(0.0), -1, 0, []
        for r in range(n_per):
            cdf = np.float32(cdf + win[b, r])
            if win[b, r] > 0:
                last_pos = r
            if r < n_per - 1:
                bounds.append(cdf)
            if pick < 0 and u < cdf:
                pick = r
        if pick < 0:
            pick = last_pos
        idx[b] = k * n_per + pick
        margin[b] = min([abs(float(u) - float(c)) for c in bounds], default=float("inf"))
    return torch.from_numpy(idx), torch.from_numpy(margin)


@torch.no_grad()
def attention_forward(sd, which, name, query, ref, use_tanh, C):
    """Attention.forward (modelPN.py:92-122) of ``actor.<which>`` ('pointer' | 'glimpse'): -> (ref' [B,H,L], logits [B,L]).
    ref' is what the glimpse loop multiplies the softmax with (:209): W_ref(ref) for 'Bahdanau', ref itself for 'Dot'."""
    if name == "Bahdanau":
        refp = ref.permute(0, 2, 1)                                                          # :104
        q = F.linear(query, sd[f"actor.{which}.W_query.weight"], sd[f"actor.{which}.W_query.bias"]).unsqueeze(2)   # :105
        refp = F.conv1d(refp, sd[f"actor.{which}.W_ref.weight"], sd[f"actor.{which}.W_ref.bias"])                   # :106
        expanded = q.repeat(1, 1, refp.shape[2])                                             # :107
        V = sd[f"actor.{which}.V"].unsqueeze(0).unsqueeze(0).repeat(ref.shape[0], 1, 1)      # :108
        logits = torch.bmm(V, torch.tanh(expanded + refp)).squeeze(1)                        # :109
    elif name == "Dot":
        logits = torch.bmm(ref, query.unsqueeze(2)).squeeze(2)                               # :112-113
        refp = ref.permute(0, 2, 1)                                                          # :114
    else:
        raise NotImplementedError(name)                                                      # :116-117
    if use_tanh:
        logits = C * torch.tanh(logits)                                                      # :119-120
    return refp, logits


def pointer_forward(sd, inputs, n_cat, n_per, latent=None, C=10.0, use_tanh=True, sample_seed=None, draw_margins=None,
                    attention="Dot", n_glimpses=0):
    """PointerNet.forward (modelPN.py:175-241); greedy, or with ``sample_seed`` the sampling mode (:227-228) with the draws
    taken from the counter-based stream (multinomial_from_stream).  ``attention`` 'Dot' | 'Bahdanau' (:80-90, 103-114),
    ``n_glimpses`` glimpse rounds per step (:208-211; the reference's configs run 'Dot' with 0).

    inputs  [B, L, 8] fp32 with L = n_cat * n_per (assert at modelPN.py:182)
    latent  None (Low net) or list of n_cat [B, L] tensors = the Low net's returned logits
    returns (probs list T x [B,L], idxs list T x [B] int64, logits list T x [B,L]).
    ``logits`` carries -inf at every previously selected position (in-place mask,
    modelPN.py:165-173,214) but NOT the latent bias nor the window mask (modelPN.py:239).
    """
    B, L, _ = inputs.shape
    assert L == n_cat * n_per
    hidden = sd["actor.decoder_start_input"].numel()
    enc = _lstm_module(sd, "encoder", hidden)
    dec = _lstm_module(sd, "decoder", hidden)

    if "actor.embedding1.weight" in sd:                                                      # :183-188 (embedding_size != 0)
        x1 = F.embedding(inputs[:, :, 0].long(), sd["actor.embedding1.weight"])
        embedded = torch.cat((x1, inputs[:, :, 1:]), 2)
    else:
        embedded = inputs.clone()
    embedded = F.linear(embedded, sd["actor.embedding2.weight"], sd["actor.embedding2.bias"])  # :190
    enc_out, (h, c) = enc(embedded)                                                        # :191

    chosen = torch.zeros(B, L, dtype=torch.bool)                                           # :196
    rows = torch.arange(B)
    idx = None
    x = sd["actor.decoder_start_input"].unsqueeze(0).repeat(B, 1)                          # :202
    all_probs, all_idx, all_logits = [], [], []
    for k in range(n_cat):                                                                  # :204
        _, (h, c) = dec(x.unsqueeze(1), (h, c))                                             # :205
        query = h.squeeze(0)                                                                # :207
        if idx is not None:                                                                 # :169-172 (the mask of this step)
            chosen[rows, idx] = True
        for _ in range(n_glimpses):                                                         # :208-211
            refp, glogits = attention_forward(sd, "glimpse", attention, query, enc_out, False, C)
            glogits[chosen] = NEG_INF
            query = torch.bmm(refp, F.softmax(glogits, dim=1).unsqueeze(2)).squeeze(2)
        _, logits = attention_forward(sd, "pointer", attention, query, enc_out, use_tanh, C)   # :213
        logits[chosen] = NEG_INF                                                            # :214
        biased = logits + latent[k] if latent else logits.clone()                           # :215-218
        biased[:, : k * n_per] = NEG_INF                                                    # :220-222
        biased[:, (k + 1) * n_per:] = NEG_INF
        probs = F.softmax(biased, dim=1)                                                    # :224
        if sample_seed is None:
            _, idx = torch.max(probs, dim=1)                                                # :226 (first max wins)
        else:
            idx, m = multinomial_from_stream(probs, k, n_cat, n_per, sample_seed)           # :228; the re-draw of :229-234
            if draw_margins is not None:                                                    # cannot trigger: windows are disjoint
                draw_margins.append(m)
        x = embedded[rows, idx, :]                                                          # :235
        all_probs.append(probs)
        all_idx.append(idx)
        all_logits.append(logits)
    return all_probs, all_idx, all_logits


def qos_calc(rows_qos, cons):
    """calc (modelPN.py:15-32) for ONE problem.

    rows_qos  float32 [T,4] = the selected candidates' (q0,q1,q2,q3)
    cons      [(lo0,hi0),(lo1,hi1)] python floats read from the step-0 action row (:51-54)
    returns (violate int, objFunc float)
    """
    ind = [np.asarray(rows_qos[:, j], dtype=np.float32) for j in range(QOS_NUM)]           # :19
    violate = 0
    for i in range(CONS_NUM):
        prod = np.cumprod(ind[i + 2])[-1]                                                   # :20
        lo, hi = cons[i]
        if prod < lo or prod > hi:                                                          # :23
            violate += 1
    n_real = int((ind[0] > 0).sum())                                                        # :26-28
    with np.errstate(divide="ignore", invalid="ignore"):   # n_real == 0 -> nan/inf, as numpy does
        obj = (np.sum(ind[0]) / n_real + 1 - np.min(ind[1])) / 2                            # :29
    return violate, float(obj)


def reward(actions, level="High", tag=0):
    """reward (modelPN.py:35-72) without the print (:67).  actions: list T x [B,8]."""
    B = actions[0].shape[0]
    act = torch.stack(actions, dim=1).numpy()                                               # [B,T,8]
    out = []
    for b in range(B):
        first = act[b, 0]
        cons = [(float(first[tag + QOS_NUM + 2 * kk]), float(first[tag + 1 + QOS_NUM + 2 * kk]))
                for kk in range(CONS_NUM)]                                                  # :51-54
        violate, obj = qos_calc(act[b, :, tag: tag + QOS_NUM], cons)
        out.append(violate if level == "Low" else round(violate + obj, 5))                  # :58-61
    return torch.FloatTensor(out)                                                           # :68


@torch.no_grad()
def combinatorial_forward(sd, inputs, n_cat, n_per, latent=None, level="Low", training="RL",
                          C=10.0, use_tanh=True, sample_seed=None, draw_margins=None, attention="Dot", n_glimpses=0):
    """CombinatorialRL.forward (modelPN.py:282-306), sample='greedy' (or sampled from the stream of ``sample_seed``)."""
    B = inputs.shape[0]
    probs, idxs, logits = pointer_forward(sd, inputs, n_cat, n_per, latent, C, use_tanh, sample_seed, draw_margins,
                                          attention, n_glimpses)
    rows = torch.arange(B)
    actions = [inputs[rows, i, :] for i in idxs]                                            # :293-295
    action_probs = [p[rows, i] for p, i in zip(probs, idxs)]                                # :297-299
    if training == "RL":
        R = reward(actions, level=level, tag=1 if "actor.embedding1.weight" in sd else 0)   # :301-304, tag: :42-45
        return R, action_probs, actions, idxs, logits
    return probs, action_probs, actions, idxs, logits


@torch.no_grad()
def two_level_greedy(sd_low, sd_high, inputs, n_cat, n_per, C=10.0, use_tanh=True, sample_high_seed=None,
                     attention="Dot", n_glimpses=0):
    """The eval block of trainPNHigh.py:138-139: Low greedy ("SL") -> latent -> High greedy ("RL").

    Returns dict with idx_low/idx_high [B,T] int64, R [B] fp32, actions [B,T,8], action_probs [B,T],
    win_low/win_high [B,T,K] (in-window logits: Low's raw, High's after adding Low's), margin
    [B,T] (top-1 minus top-2 of the High decision; +inf when K == 1).
    """
    B, L, _ = inputs.shape
    _, _, _, idx_low, latent = combinatorial_forward(sd_low, inputs, n_cat, n_per, None, "Low", "SL",
                                                     C, use_tanh, attention=attention, n_glimpses=n_glimpses)
    draws = [] if sample_high_seed is not None else None      # trainPNHigh.py:83-84: Low greedy -> latent, High sampled
    R, aprob, actions, idx_high, logits_high = combinatorial_forward(
        sd_high, inputs, n_cat, n_per, latent, "High", "RL", C, use_tanh, sample_high_seed, draws, attention, n_glimpses)
    win_low = torch.stack([latent[k][:, k * n_per:(k + 1) * n_per] for k in range(n_cat)], 1)
    win_high = torch.stack([(logits_high[k] + latent[k])[:, k * n_per:(k + 1) * n_per]
                            for k in range(n_cat)], 1)
    return {
        "idx_low": torch.stack(idx_low, 1), "idx_high": torch.stack(idx_high, 1), "R": R,
        "actions": torch.stack(actions, 1), "action_probs": torch.stack(aprob, 1),
        "win_low": win_low, "win_high": win_high,
        "margin_low": decision_margin(win_low, inputs),
        "margin_high": decision_margin(win_high, inputs) if draws is None else torch.stack(draws, 1),
        "latent": latent,
    }


def decision_margin(win, inputs):
    """How far each decision is from flipping to a DIFFERENT candidate: top-1 logit minus the best
    logit among window rows whose input row differs from the top-1's row (+inf when there is none).
    Rows that are identical (dummy rows of an absent category, loadData.py:148, or the cyclic
    padding of :137-138) are the same selection: whichever copy wins, the action row and the next
    decoder input (modelPN.py:235) are identical, so a flip among them changes nothing downstream.
    win [B,T,K], inputs [B,T*K,F] -> [B,T]."""
    B, T, K = win.shape
    rows = inputs.view(B, T, K, -1)
    top = win.argmax(dim=2, keepdim=True)                                   # first max
    top_row = torch.gather(rows, 2, top.unsqueeze(-1).expand(B, T, 1, rows.shape[-1]))
    other = (rows != top_row).any(dim=-1)                                   # [B,T,K]
    rival = torch.where(other, win, torch.full_like(win, NEG_INF)).max(dim=2).values
    return torch.gather(win, 2, top).squeeze(2) - rival
