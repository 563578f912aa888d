"""Pointer-network side of the ML+2PN path, behind the reference's entry points
(/root/reference/src/models/modelPN.py): ``CombinatorialRL`` (:244-306), ``PointerNet``
(:126-241), ``reward`` (:35-72), ``calc`` (:15-32).  Same constructor arguments, same
``state_dict`` keys/shapes (``actor.embedding2.*``, ``actor.encoder.*_l0``, ``actor.decoder.*_l0``,
``actor.decoder_start_input``), same 5-tuple from ``forward`` — but every tensor operation runs
in the hand-written gfx950 kernels of libgnnpn_hip.so (``ops.py``); torch modules are used here
only as parameter containers.

Scope of this build = the configuration the reference ships
(/root/reference/environment.ini:49-79, src/models/trainPNHigh.py:208-236):
``embedding_size=0``, ``n_glimpses=0``, ``attention='Dot'``; ``sample='greedy'`` (the inference path) and the forward of
the sampling mode (``sample='sample'``: every pick drawn from the window softmax, modelPN.py:227-228 — SURVEY.md §8f
row 3; its REINFORCE step is trainPNHigh.py here).  The attention forms those configurations leave switched off —
``attention='Bahdanau'`` (modelPN.py:80-90,103-109) and ``n_glimpses > 0`` (:208-211), SURVEY.md §8f row 4 — decode through
the general kernel (gnnpn_pointer_decode_attn_f32: one net per call, greedy or sampled).  ``embedding_size != 0``
(embeddingTag=1: rows [category | 8 floats], modelPN.py:153-154,183-188) decodes one net per call as well; its sampled form takes
the streaming decode kernel.  Those forms run in fp32 (the exact-split builds are the shipped configuration's).
"""
import math

import os

import torch
from torch import nn

from . import custom_ops, ops

# Fold embedding2 (modelPN.py:190) and the encoder LSTM's input projection into ONE [4H, 8] matrix
# evaluated inside the recurrent kernel (W_ih.(W_e x + b_e) + b_ih = (W_ih W_e) x + (W_ih b_e + b_ih)):
# an exact algebraic identity that removes the [B*L,256]x[256,1024] GEMM and the 246 MB pre-gate
# round trip per net.  It rounds differently from the reference's two-stage evaluation (by about the
# reference's own rounding error); set False for the literal two-stage order.
FOLD_INPUT_PROJECTION = True

qosandcons = 8   # modelPN.py:10
qosNum = 4       # modelPN.py:11
consNum = 2      # modelPN.py:12


class LatentWindows:
    """What a Low-level forward hands to the High-level forward (`latent`, trainPNHigh.py:138-139).

    The reference passes a python list of T full-length ``[B, L]`` logits tensors although only
    the ``[k*K,(k+1)*K)`` window of entry k can influence anything (modelPN.py:216,220-222).
    This object keeps the compact ``win [B,T,K]`` tensor the kernels exchange and behaves like
    that list on demand: ``len()``, iteration and ``[k]`` materialise the reference's full-length
    tensor for step k (window logits in place, -inf at previously chosen positions, the remaining
    positions computed by gnnpn_attention_logits_f32)."""

    def __init__(self, win, idx, enc_out, queries, tanh_c, use_tanh, bahdanau=None):
        self.win, self.idx = win, idx
        self._enc_out, self._queries = enc_out, queries            # queries: what the pointer attention saw (after the glimpse rounds)
        self._tanh_c, self._use_tanh = tanh_c, use_tanh
        self._bahdanau, self._ref = bahdanau, None                 # 'Bahdanau': the pointer module's {wq, bq, wref, bref, v} (modelPN.py:82-90)

    def __len__(self):
        return self.win.shape[1]

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        if k < 0:
            k += len(self)
        if self._bahdanau is not None:                             # V . tanh(W_query q + b + W_ref(enc) + b) (modelPN.py:103-109)
            from . import ops
            a = {n: v.detach().float().contiguous() for n, v in self._bahdanau.items()}
            B, L, H = self._enc_out.shape
            if self._ref is None:                                  # once per forward
                self._ref = ops.linear(self._enc_out.reshape(B * L, H), a["wref"].reshape(H, H).contiguous(), a["bref"]).view(B, L, H)
            qp = ops.linear(self._queries[:, k, :].contiguous(), a["wq"], a["bq"])
            return ops.attention_logits_bahdanau(self._ref, qp, a["v"], k, self.idx, self._tanh_c, self._use_tanh)
        return torch.ops.gnnpn.attention_logits(self._enc_out, self._queries, k, self.idx, self._tanh_c, self._use_tanh)

    def __iter__(self):
        return (self[k] for k in range(len(self)))

    def copy(self):      # modelPN.py:291 calls logits.copy()
        return self


def _window_tensor(latent, n_cat, n_per):
    """Accept the reference's list-of-[B,L] form as well as LatentWindows."""
    if latent is None:
        return None
    if isinstance(latent, LatentWindows):
        return latent.win
    if len(latent) != n_cat:
        raise ValueError(f"latent has {len(latent)} steps, expected {n_cat}")
    return torch.stack([latent[k][:, k * n_per:(k + 1) * n_per] for k in range(n_cat)], 1).contiguous()


class Attention(nn.Module):
    """Parameter container of Attention (modelPN.py:75-90): same names and shapes (``W_query`` Linear, ``W_ref`` 1x1
    Conv1d, ``V``); 'Dot' has no parameters.  The arithmetic (:92-122) runs in gnnpn_pointer_decode_attn_f32."""

    def __init__(self, hidden_size, use_tanh=False, C=10, name="Bahdanau", use_cuda=True):
        super().__init__()
        if name not in ops.ATTENTION_NAMES:
            raise NotImplementedError(name)                                            # :116-117
        self.use_tanh, self.C, self.name = use_tanh, C, name
        if name == "Bahdanau":
            self.W_query = nn.Linear(hidden_size, hidden_size)                         # :83
            self.W_ref = nn.Conv1d(hidden_size, hidden_size, 1, 1)                     # :84
            self.V = nn.Parameter(torch.FloatTensor(hidden_size))                      # :86-90
            self.V.data.uniform_(-(1. / math.sqrt(hidden_size)), 1. / math.sqrt(hidden_size))

    def side(self):
        if self.name != "Bahdanau":
            return None
        return {"wq": self.W_query.weight, "bq": self.W_query.bias, "wref": self.W_ref.weight, "bref": self.W_ref.bias,
                "v": self.V}


class PointerNet(nn.Module):
    """Parameter container + kernel driver for PointerNet (modelPN.py:126-241)."""

    def __init__(self, embedding_size, hidden_size, seq_len, n_glimpses, tanh_exploration, use_tanh, attention,
                 sNumber, sCategory, use_cuda=True, level="low", mask=False):
        super().__init__()
        if n_glimpses < 0:
            raise ValueError("n_glimpses >= 0")
        self.embedding_size, self.hidden_size, self.n_glimpses = embedding_size, hidden_size, n_glimpses
        self.attention = attention
        # general = outside the shipped configuration ('Dot', no glimpses): decoded by gnnpn_pointer_decode_attn_f32
        self.general = attention != "Dot" or n_glimpses != 0
        self.seq_len, self.use_cuda, self.level = seq_len, use_cuda, level
        self.serNumber, self.serCategory = sNumber, sCategory
        self.C, self.use_tanh, self.mask = float(tanh_exploration), bool(use_tanh), mask
        if embedding_size != 0:                                                        # :153-154 (embeddingTag=1: rows [cat | 8 floats])
            self.embedding1 = nn.Embedding(sCategory, embedding_size)
        self.embedding2 = nn.Linear(embedding_size + qosandcons, hidden_size)          # :155
        self.encoder = nn.LSTM(hidden_size, hidden_size, batch_first=True)             # :157 (container)
        self.decoder = nn.LSTM(hidden_size, hidden_size, batch_first=True)             # :158 (container)
        self.pointer = Attention(hidden_size, use_tanh=use_tanh, C=tanh_exploration, name=attention, use_cuda=use_cuda)   # :159
        self.glimpse = Attention(hidden_size, use_tanh=False, name=attention, use_cuda=use_cuda)                           # :160
        self.decoder_start_input = nn.Parameter(torch.FloatTensor(hidden_size))        # :162-163
        self.decoder_start_input.data.uniform_(-(1. / math.sqrt(hidden_size)), 1. / math.sqrt(hidden_size))
        self._packed = None
        self.fold_on_device = False
        # sampling mode: the reference draws from torch's global generator; here the draws of forward call n come from
        # the counter-based stream seeded (sample_seed, n) — reproducible, and a function of nothing but the two numbers
        self.sample_seed = int(torch.initial_seed()) & 0x7FFFFFFF
        self.sample_calls = 0

    def next_sample_seed(self):
        self.sample_calls += 1
        return (self.sample_seed * 1000003 + self.sample_calls) & 0x7FFFFFFFFFFFFFFF

    # weights are re-laid-out once (k-major float4 packing for the recurrent kernels)
    def _load_from_state_dict(self, *a, **k):
        self._packed = None
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def packed(self):
        if self._packed is None:
            f = lambda t: t.detach().float().contiguous()   # noqa: E731
            self._packed = {
                "emb_w": f(self.embedding2.weight), "emb_b": f(self.embedding2.bias),
                "enc_wih": f(self.encoder.weight_ih_l0), "enc_bih": f(self.encoder.bias_ih_l0),
                "enc_whh": ops.pack_lstm_weight(f(self.encoder.weight_hh_l0)), "enc_bhh": f(self.encoder.bias_hh_l0),
                "dec_wih": ops.pack_lstm_weight(f(self.decoder.weight_ih_l0)), "dec_bih": f(self.decoder.bias_ih_l0),
                "dec_whh": ops.pack_lstm_weight(f(self.decoder.weight_hh_l0)), "dec_bhh": f(self.decoder.bias_hh_l0),
                "start": f(self.decoder_start_input),
            }
            # folded input projection, formed in fp64 and rounded once.  Weights loaded for inference fold on the host (the
            # arithmetic every committed parity record was measured with); while a trainer steps the weights
            # (``fold_on_device``, set by trainPNHigh.ActorAdam) the same fp64 products run on the device instead of a
            # 1 MB round trip through the host per step.
            # the exact split of the two recurrent matrices, once per set of weights (ops.pack_lstm_split_weights: what a "split"
            # launch otherwise works out for itself every time); not while a capture is recording (an allocation there would live in
            # the graph's pool) — the kernels then split for themselves, same bits
            if self.hidden_size == 256 and self.embedding2.weight.is_cuda and not torch.cuda.is_current_stream_capturing():
                self._packed["enc_whh_split"] = ops.pack_lstm_split_weights(self._packed["enc_whh"])
                self._packed["dec_whh_split"] = ops.pack_lstm_split_weights(self._packed["dec_whh"])
            if self.embedding_size != 0:     # the folded input side needs W_e [H, 8]: with the category embedding the literal order runs
                self._packed["whh_absmax"] = None
                return self._packed
            where = (lambda t: t.detach().double()) if self.fold_on_device else (lambda t: t.detach().double().cpu())
            dev = self.embedding2.weight.device
            back = lambda t: t.float().contiguous().to(dev)   # noqa: E731
            w_ih, w_e, b_e = where(self.encoder.weight_ih_l0), where(self.embedding2.weight), where(self.embedding2.bias)
            self._packed["enc_wfold"] = back(w_ih @ w_e)
            self._packed["enc_bfold"] = back(w_ih @ b_e + where(self.encoder.bias_ih_l0))
            # the same for the decoder cell's input side, plus its step-0 gates W_ih.start + b_ih
            d_ih, d_b = where(self.decoder.weight_ih_l0), where(self.decoder.bias_ih_l0)
            self._packed["dec_wfold"] = back(d_ih @ w_e)
            self._packed["dec_bfold"] = back(d_ih @ b_e + d_b)
            self._packed["dec_sfold"] = back(d_ih @ where(self.decoder_start_input) + d_b)
            self._packed["whh_absmax"] = None      # on demand (check_precision): it costs two device-to-host reads
        return self._packed

    def check_precision(self, precision):
        """"f16" holds W_hh as plain fp16: refuse weights outside fp16's finite range.  The exact split ("split") scales every
        gate column by its own power of two (csrc/coop_common.h) and takes any finite weight, as fp32 does."""
        if precision == "f16":
            w = self.packed()
            if w["whh_absmax"] is None:          # largest |W_hh|
                w["whh_absmax"] = float(max(self.encoder.weight_hh_l0.detach().abs().max(),
                                            self.decoder.weight_hh_l0.detach().abs().max()))
            m = w["whh_absmax"]
            if not m < 6.0e4:
                raise ops.GnnpnError(f"precision={precision!r}: max |W_hh| = {m:g} does not fit fp16 operands; use 'f32' or 'split'")

    def encode_args(self, inputs, fold=None):
        """One entry of the ``nets`` list of ops.lstm_encode (+ the embedded tensor when the literal
        two-stage order is asked for: embedding2 (:190), then x_t.W_ih^T + b_ih (inside :191))."""
        B, L, F = inputs.shape
        assert L == self.seq_len                                                        # :182
        w = self.packed()
        fold = FOLD_INPUT_PROJECTION if fold is None else fold
        if self.embedding_size != 0:
            # :183-190 — x1 = embedding1(inputs[:, :, 0]); embedded = embedding2(cat(x1, inputs[:, :, 1:])): the lookup + concat
            # kernel of the GNN front end (gnnpn_embed_concat_f32), then the literal two-stage order (the folded [4H, 8] matrix
            # has no place for a per-row table entry)
            if F != 1 + qosandcons:
                raise ops.GnnpnError(f"embedding_size != 0: rows [category | {qosandcons} floats] expected, got {F} columns")
            flat = torch.ops.gnnpn.embed_concat(inputs.reshape(B * L, F), self.embedding1.weight.detach().float().contiguous())
            embedded = torch.ops.gnnpn.linear(flat, w["emb_w"], w["emb_b"])
            pregates = torch.ops.gnnpn.linear(embedded, w["enc_wih"], w["enc_bih"])
            return {"pregates": pregates.view(B, L, 4 * self.hidden_size), "whh": w["enc_whh"], "bhh": w["enc_bhh"],
                    "whh_split": w.get("enc_whh_split")}, embedded.view(B, L, self.hidden_size)
        if fold:
            return {"inputs": inputs, "w_in": w["enc_wfold"], "b_in": w["enc_bfold"], "whh": w["enc_whh"],
                    "bhh": w["enc_bhh"], "whh_split": w.get("enc_whh_split")}, None
        embedded = torch.ops.gnnpn.linear(inputs.reshape(B * L, F), w["emb_w"], w["emb_b"])
        pregates = torch.ops.gnnpn.linear(embedded, w["enc_wih"], w["enc_bih"])
        H = self.hidden_size
        return {"pregates": pregates.view(B, L, 4 * H), "whh": w["enc_whh"], "bhh": w["enc_bhh"], "whh_split": w.get("enc_whh_split")}, \
            embedded.view(B, L, H)

    def decode_args(self, embedded, enc_out, h_n, c_n, latent_win=None, latent_from=-1, fold=None, sample_seed=None):
        """One entry of the ``nets`` list of ops.pointer_decode (embedded=None: picks are embedded
        in-kernel from the raw rows; fold: the cell's input side uses the folded [4H,8] matrix)."""
        w = self.packed()
        fold = FOLD_INPUT_PROJECTION if fold is None else fold
        d = {"embedded": embedded, "emb_w": w["emb_w"], "emb_b": w["emb_b"], "enc_out": enc_out, "h0": h_n,
             "c0": c_n, "start": w["start"], "wih": w["dec_wih"], "whh": w["dec_whh"], "bih": w["dec_bih"],
             "bhh": w["dec_bhh"], "latent_win": latent_win, "latent_from": latent_from, "whh_split": w.get("dec_whh_split")}
        if fold and embedded is None:
            d.update(xw_fold=w["dec_wfold"], xb_fold=w["dec_bfold"], start_fold=w["dec_sfold"])
        if sample_seed is not None:
            d.update(sample=True, sample_seed=int(sample_seed))
        return d

    @torch.no_grad()
    def run(self, inputs, latent=None, want_queries=False, fold=None, sample_seed=None, encoder_precision="f32"):
        """Encode + decode (greedy, or with ``sample_seed`` every pick drawn from the window softmax); returns the decode
        dict of ops.pointer_decode plus enc_out.  ``encoder_precision``: operand arithmetic of the ENCODER's recurrent product
        ("f32" | "split" | "f16": the cooperative encoder, H = 256) — the decode kernels this single-net entry reaches (the general
        attention kernel, the literal two-stage cooperative build, the streaming form) compute in fp32."""
        inputs = inputs.contiguous()
        enc_args, embedded = self.encode_args(inputs, fold)
        enc, h_n, c_n = custom_ops.lstm_encode([enc_args], precision=encoder_precision)
        impl = 0
        if self.embedding_size != 0:          # the decode kernels gather 8-feature action rows: the category column rejoins below
            rows9, inputs = inputs, inputs[:, :, 1:].contiguous()
            fold = False
            if sample_seed is not None:       # the cooperative build that draws is the folded one: sampled decodes of the embedding
                impl = 1                      # form take the per-workgroup streaming kernel (same draws, same arithmetic)
        if self.general:                      # (greedy or drawn: gnnpn_pointer_decode_attn_f32 takes the same sample / seed fields)
            out = ops.pointer_decode_attn(
                self.decode_args(embedded, enc[0], h_n[0], c_n[0], _window_tensor(latent, self.serCategory, self.serNumber),
                                 fold=False, sample_seed=sample_seed),
                inputs, self.serCategory, self.serNumber, self.attention, self.n_glimpses, self.pointer.side(),
                self.glimpse.side(), self.C, self.use_tanh, want_queries)
            out["enc_out"] = enc[0]
            return self._with_category(out, rows9) if self.embedding_size != 0 else out
        out = custom_ops.pointer_decode(
            [self.decode_args(embedded, enc[0], h_n[0], c_n[0],
                              _window_tensor(latent, self.serCategory, self.serNumber), fold=fold, sample_seed=sample_seed)],
            inputs, self.serCategory, self.serNumber, self.C, self.use_tanh, want_queries, impl=impl)[0]
        out["enc_out"] = enc[0]
        return self._with_category(out, rows9) if self.embedding_size != 0 else out

    @staticmethod
    def _with_category(out, rows9):
        """actions = inputs[b, idx, :] (modelPN.py:293-295) keeps the category column when the rows carry one."""
        cat = torch.gather(rows9[:, :, 0], 1, out["idx"].long()).unsqueeze(-1)
        out["actions"] = torch.cat([cat, out["actions"]], dim=2)
        return out

    def forward(self, inputs, latent, sample="sample"):
        """(probs, idxs, logits) lists as PointerNet.forward returns them (:241).  probs / logits
        are lazily materialised full-length views (see LatentWindows)."""
        out = self.run(inputs, latent, want_queries=True,
                       sample_seed=None if sample == "greedy" else self.next_sample_seed())      # :225-228
        lat = LatentWindows(out["win_logits"], out["idx"], out["enc_out"], out["queries"], self.C, self.use_tanh, self.pointer.side())
        idxs = [out["idx"][:, k].long() for k in range(self.serCategory)]
        return _ProbList(out, latent, self.serCategory, self.serNumber), idxs, lat


class _ProbList:
    """The reference's ``probs`` list (T x [B,L] softmax rows, zero outside the step's window)."""

    def __init__(self, out, latent, n_cat, n_per):
        win = out["win_logits"]
        lw = _window_tensor(latent, n_cat, n_per)
        self._p = torch.softmax(win if lw is None else win + lw, dim=2)   # [B,T,K], compat path only
        self._n_per, self._L = n_per, n_cat * n_per

    def __len__(self):
        return self._p.shape[1]

    def __getitem__(self, k):
        B = self._p.shape[0]
        full = torch.zeros((B, self._L), dtype=self._p.dtype, device=self._p.device)
        full[:, k * self._n_per:(k + 1) * self._n_per] = self._p[:, k]
        return full

    def __iter__(self):
        return (self[k] for k in range(len(self)))


def calc(services, constraints, sCategory):
    """calc (modelPN.py:15-32) for ONE problem given as the reference passes it (list of T rows of
    >= 4 values, constraints [[lo,hi]] per global constraint).  Host helper kept for API parity;
    the batched device version is ops.qos_reward."""
    rows = torch.stack([torch.as_tensor(s)[:qosNum].float().cpu() for s in services]).unsqueeze(0)
    act = torch.zeros((1, rows.shape[1], 8), dtype=torch.float32)
    act[0, :, :qosNum] = rows[0]
    for i in range(consNum):
        act[0, 0, qosNum + 2 * i], act[0, 0, qosNum + 2 * i + 1] = constraints[i][0][-2], constraints[i][0][-1]
    dev = torch.device("cuda")
    violate = int(torch.ops.gnnpn.qos_reward(act.to(dev), 0).item())
    total = float(torch.ops.gnnpn.qos_reward(act.to(dev), 1).item())
    return violate, total - violate, []


def reward(sample_solution, optSolutions, sCategory, USE_CUDA=False, level="Low", embedding_size=20,
           verbose=False):
    """reward (modelPN.py:35-72): list of T ``[B, 8]`` action tensors -> FloatTensor [B].
    The reference prints the whole list (:67); here only with ``verbose=True``."""
    actions = torch.stack(list(sample_solution), dim=1)
    if actions.shape[-1] == 1 + qosandcons:          # embedding_size != 0: tag = 1 (modelPN.py:42-45), the category column in front
        actions = actions[..., 1:]
    actions = actions.contiguous()
    R = torch.ops.gnnpn.qos_reward(actions, 0 if level == "Low" else 1)
    if verbose:
        lst = R.tolist()
        print(f"{level}, {sum(1 for v in lst if v >= 1)}, {sum(lst) / max(len(lst), 1)}: ", lst)
    return R


class CombinatorialRL(nn.Module):
    """CombinatorialRL (modelPN.py:244-306)."""

    def __init__(self, embedding_size, hidden_size, seq_len, n_glimpses, tanh_exploration, use_tanh, reward,
                 attention, sNumber, sCategory, use_cuda=True, level="Low", mask=False):
        super().__init__()
        self.reward = reward
        self.use_cuda, self.level, self.embedding_size = use_cuda, level, embedding_size
        self.sNumber, self.serCategory = sNumber, sCategory
        self.actor = PointerNet(embedding_size, hidden_size, seq_len, n_glimpses, tanh_exploration, use_tanh,
                                attention, sNumber, sCategory, use_cuda, level=level, mask=mask)

    @torch.no_grad()
    def forward(self, inputs, labs, latent=None, sample="sample", training="RL"):
        """-> (R | probs, action_probs T x [B], actions T x [B,8], action_idxs T x [B] int64, latent_p)."""
        out = self.actor.run(inputs, latent, want_queries=True,
                             sample_seed=None if sample == "greedy" else self.actor.next_sample_seed())   # modelPN.py:225-228
        T = self.serCategory
        action_idxs = [out["idx"][:, k].long() for k in range(T)]
        actions = [out["actions"][:, k, :] for k in range(T)]                       # :293-295
        action_probs = [out["pick_prob"][:, k] for k in range(T)]                   # :297-299
        latent_p = LatentWindows(out["win_logits"], out["idx"], out["enc_out"], out["queries"], self.actor.C,
                                 self.actor.use_tanh, self.actor.pointer.side())
        if training == "RL":                                                        # :301-304
            R = self.reward(actions, labs, self.serCategory, USE_CUDA=self.use_cuda, level=self.level,
                            embedding_size=self.embedding_size)
            return R, action_probs, actions, action_idxs, latent_p
        return (_ProbList(out, latent, T, self.sNumber), action_probs, actions, action_idxs, latent_p)


@torch.no_grad()
def default_precision(low, high, fold=None, sampling=False, decode_impl=0):
    """The arithmetic of the two recurrent W_hh.h products when the caller does not choose: the exact three-piece split
    ("split": fp32 in, fp32 out, no operand bit dropped, error bound and measured error below the fp32 matrix chain's,
    identical selections on every pinned problem, 1.4-1.5 x the throughput — DESIGN.md section 5; profiles/LOG_r01_r04.md section 12, and what bench.py
    measures) wherever its kernels apply — the cooperative, folded, greedy form: hidden size 256, windows of at most 16
    candidates, no category embedding, no general attention — and the fp32 matrix cores ("f32") everywhere else."""
    la, ha = low.actor, high.actor
    fits = all(a.hidden_size == 256 and a.serNumber <= 16 and a.embedding_size == 0 and not a.general for a in (la, ha))
    folded = FOLD_INPUT_PROJECTION if fold is None else bool(fold)
    return "split" if fits and folded and not sampling and decode_impl != 1 else "f32"      # (decode_impl 1: the streaming form)


def two_level_greedy(low, high, inputs, fold=None, precision=None, decode_impl=0, lds_kb=0, write_through=False, ws=None,
                     sample_high_seed=None, paired_start=False):
    """The inference harness of trainPNHigh.py:138-139 as one device-resident call: both encoders in
    ONE launch (they are independent), both decoders in ONE launch (High biased by Low's window
    logits, one step behind), QoS reward.  Returns dict(idx_low, idx_high [B,T] int32, R [B], actions [B,T,8],
    action_probs [B,T], win_low, win_high_raw [B,T,K]) — the High decision is taken on
    win_high_raw + win_low (modelPN.py:216).
    precision: None (default: ``default_precision`` — "split" wherever its kernels apply, else "f32") | "f32" (fp32 matrix
    cores) | "split" (both W_hh.h products from fp32 operands split EXACTLY into three fp16 pieces, six products per term
    on the fp16 matrix cores, fp32 accumulate: no operand bit dropped, error bound below the fp32 chain's, DESIGN.md
    section 12) | "f16" (encoder operands in plain fp16: opt-in reduced precision).
    sample_high_seed: the High level DRAWS its picks from that stream instead of taking the argmax — the forward of the
    PNHigh training step (trainPNHigh.py:83-84: Low greedy -> latent, High sample='sample'); the Low level stays greedy.
    decode_impl / lds_kb / write_through / paired_start / ws: per-call launch options of the two recurrent kernels (ops.lstm_encode,
    ops.pointer_decode): which decoder build, LDS-footprint placement control, hand-off form, whose workspaces."""
    inputs = inputs.contiguous()
    la, ha = low.actor, high.actor
    if precision is None:
        precision = default_precision(low, high, fold, sample_high_seed is not None, decode_impl)
    if la.general or ha.general or la.embedding_size != 0 or ha.embedding_size != 0:
        # 'Bahdanau' attention / glimpses / the category embedding (embeddingTag=1): one net per call.  A non-fp32 ``precision``
        # applies to the L-step encoder recurrences (the cooperative encoder: "split" exact, "f16" reduced); the T decoder steps of
        # these forms run kernels that are built in fp32 only (round 5: this used to raise NotImplementedError)
        dl = la.run(inputs, None, fold=fold, encoder_precision=precision)
        dh = ha.run(inputs, LatentWindows(dl["win_logits"], dl["idx"], None, None, la.C, la.use_tanh), fold=fold,
                    sample_seed=sample_high_seed, encoder_precision=precision)     # High drawn (modelPN.py:227-228) or greedy; Low greedy
        R = torch.ops.gnnpn.qos_reward(dh["actions"][..., -qosandcons:].contiguous(), 0 if high.level == "Low" else 1)
        return {"idx_low": dl["idx"], "idx_high": dh["idx"], "R": R, "actions": dh["actions"],
                "action_probs": dh["pick_prob"], "win_low": dl["win_logits"], "win_high_raw": dh["win_logits"]}
    # the two decoders run in ONE launch with one (C, use_tanh, window shape): the reference applies each PointerNet's
    # own (modelPN.py:119-122), so nets that differ there cannot take this fused path
    for name in ("C", "use_tanh", "serNumber", "serCategory", "hidden_size", "seq_len"):
        if getattr(la, name) != getattr(ha, name):
            raise ops.GnnpnError(f"two_level_greedy: Low and High nets differ in {name} ({getattr(la, name)!r} vs "
                                 f"{getattr(ha, name)!r}); decode them with separate CombinatorialRL.forward calls")
    la.check_precision(precision)
    ha.check_precision(precision)
    enc_l, emb_l = la.encode_args(inputs, fold)
    enc_h, emb_h = ha.encode_args(inputs, fold)
    enc, h_n, c_n = custom_ops.lstm_encode([enc_l, enc_h], precision=precision, lds_kb=lds_kb, write_through=write_through, ws=ws,
                                           paired_start=paired_start)
    del enc_l, enc_h
    dl, dh = custom_ops.pointer_decode([la.decode_args(emb_l, enc[0], h_n[0], c_n[0], fold=fold),
                                 ha.decode_args(emb_h, enc[1], h_n[1], c_n[1], latent_from=0, fold=fold,
                                                sample_seed=sample_high_seed)],
                                inputs, la.serCategory, la.serNumber, la.C, la.use_tanh,
                                precision="split" if precision == "split" else "f32", impl=decode_impl, lds_kb=lds_kb,
                                write_through=write_through, ws=ws, paired_start=paired_start)
    R = torch.ops.gnnpn.qos_reward(dh["actions"], 0 if high.level == "Low" else 1)
    return {"idx_low": dl["idx"], "idx_high": dh["idx"], "R": R, "actions": dh["actions"],
            "action_probs": dh["pick_prob"], "win_low": dl["win_logits"], "win_high_raw": dh["win_logits"]}
