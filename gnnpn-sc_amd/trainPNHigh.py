"""REINFORCE training of the High-level pointer network behind the reference's driver
(/root/reference/src/models/trainPNHigh.py): ``TrainModel`` (:43-150) — same constructor arguments, ``train_and_validate``
— with the step of :81-110 running on the GPU: Low greedy -> latent, High sampled (picks drawn inside the decode kernel),
then the backward pass, gradient clipping and Adam of csrc/train.hip.

No autograd: the gradient of  mean_b( advantage_b * sum_k log p_{b,k} )  wrt every actor parameter is computed by the
hand-written backward kernels with the picks as constants; weights are updated in place.
"""
import os
import time

import torch

from . import evalPN, ops

ACTOR_KEYS = ("decoder_start_input", "embedding2.weight", "embedding2.bias",
              "encoder.weight_ih_l0", "encoder.weight_hh_l0", "encoder.bias_ih_l0", "encoder.bias_hh_l0",
              "decoder.weight_ih_l0", "decoder.weight_hh_l0", "decoder.bias_ih_l0", "decoder.bias_hh_l0")


def _params(actor):
    return {k: p for k, p in actor.named_parameters()}


@torch.no_grad()
def actor_gradients(actor, inputs, idx, latent_win, gscale):
    """Gradient of sum_b gscale[b] * sum_k log p_{b,k}(picks) wrt every parameter of ``actor`` (a modelPN.PointerNet):
    -> (dict name -> gradient tensor, logp [B,T]).  inputs [B,L,8]; idx [B,T] int32 picks (global positions); latent_win
    [B,T,K] the Low net's window logits (constants) or None."""
    general = bool(getattr(actor, "general", False))      # 'Bahdanau' attention / glimpse rounds (modelPN.py:83-109,208-211): csrc/train_attn.hip
    p = {k: v.detach().float().contiguous() for k, v in _params(actor).items()}
    B, L, F = inputs.shape
    H, T, K = actor.hidden_size, actor.serCategory, actor.serNumber
    flat = inputs.reshape(B * L, F).contiguous()
    E = int(getattr(actor, "embedding_size", 0))
    rows9 = None
    if E:                                           # embeddingTag=1: rows [category | 8 floats] (modelPN.py:183-188)
        if F != 9:
            raise ops.GnnpnError(f"embedding_size != 0: rows [category | 8 floats] expected, got {F} columns")
        rows9 = flat
        flat = ops.embed_concat(rows9, p["embedding1.weight"])                                         # [B L, E + 8]
    embedded = ops.linear(flat, p["embedding2.weight"], p["embedding2.bias"])                          # modelPN.py:190
    pregates = ops.linear(embedded, p["encoder.weight_ih_l0"], p["encoder.bias_ih_l0"]).view(B, L, 4 * H)
    tr = lambda w: w.t().contiguous()                                                                  # noqa: E731  [4H,H] -> k-major [H,4H]
    enc_out, gates_e, c_e = ops.lstm_train_forward(pregates, tr(p["encoder.weight_hh_l0"]), p["encoder.bias_hh_l0"])   # :191
    h0, c0 = enc_out[:, L - 1].contiguous(), c_e[:, L - 1].contiguous()
    g = {}
    if general:
        d = ops.decode_attn_train_forward(embedded.view(B, L, H), enc_out, h0, c0, p["decoder_start_input"],
                                          tr(p["decoder.weight_ih_l0"]), tr(p["decoder.weight_hh_l0"]), p["decoder.bias_ih_l0"],
                                          p["decoder.bias_hh_l0"], latent_win, idx, T, K, actor.attention, actor.n_glimpses,
                                          actor.pointer.side(), actor.glimpse.side(), actor.C, actor.use_tanh)   # :204-239
        d.update(wih=p["decoder.weight_ih_l0"], whh=p["decoder.weight_hh_l0"])                         # the backward's layout
        d_enc_out, dg_d, dx, dh0, dc0 = ops.decode_attn_train_backward(d, gscale)
        G = actor.n_glimpses
        if d["bahdanau"]:                                  # the two Attention modules' parameters (modelPN.py:82-90), d enc_out through ref
            flat_enc = enc_out.reshape(B * L, H)
            for tag, name, rounds in (("p", "pointer", 1),) + ((("g", "glimpse", G),) if G else ()):
                dqp = d[f"d_{tag}_qp"].reshape(B * T * rounds, H)
                # the queries the projections saw: q_G for the pointer, q_0 .. q_{G-1} for the glimpse rounds
                q_in = (d["q_all"][:, :, G] if tag == "p" else d["q_all"][:, :, :G]).reshape(B * T * rounds, H).contiguous()
                g[f"{name}.W_query.weight"] = ops.gemm(dqp, q_in, True, True)
                g[f"{name}.W_query.bias"] = ops.colsum(dqp)
                g[f"{name}.V"] = ops.colsum(d[f"d_{tag}_v"])
                dref = d[f"d_{tag}_ref"].view(B * L, H)
                g[f"{name}.W_ref.weight"] = ops.gemm(dref, flat_enc, True, True).view(H, H, 1)          # ref = W_ref enc + b_ref (:106)
                g[f"{name}.W_ref.bias"] = ops.colsum(dref)
                d_enc_out += ops.gemm(dref, d[f"{tag}_wref"], False, True).view(B, L, H)
            if not G:                                      # the glimpse module took no part (the reference leaves .grad = None)
                for k, v in p.items():
                    if k.startswith("glimpse."):
                        g[k] = torch.zeros_like(v)
    else:
        d = ops.decode_train_forward(embedded.view(B, L, H), enc_out, h0, c0, p["decoder_start_input"],
                                     tr(p["decoder.weight_ih_l0"]), tr(p["decoder.weight_hh_l0"]), p["decoder.bias_ih_l0"],
                                     p["decoder.bias_hh_l0"], latent_win, idx, T, K, actor.C, actor.use_tanh)   # :204-239
        d.update(wih=p["decoder.weight_ih_l0"], whh=p["decoder.weight_hh_l0"])                         # the backward's layout
        d_enc_out, dg_d, dx, dh0, dc0 = ops.decode_train_backward(d, gscale)
    dg_e = ops.lstm_train_backward(p["encoder.weight_hh_l0"], gates_e, c_e, d_enc_out, dh0, dc0)
    dgd = dg_d.view(B * T, 4 * H)
    hprev_d = torch.cat([h0.unsqueeze(1), d["h_all"][:, :-1]], 1).reshape(B * T, H).contiguous()      # h_{k-1} of every step
    g["decoder.weight_hh_l0"] = ops.gemm(dgd, hprev_d, True, True)
    g["decoder.weight_ih_l0"] = ops.gemm(dgd, d["x_all"].view(B * T, H), True, True)
    g["decoder.bias_ih_l0"] = ops.colsum(dgd)
    g["decoder.bias_hh_l0"] = g["decoder.bias_ih_l0"]
    g["decoder_start_input"] = ops.colsum(dx, rows=B, cols=H, ld=T * H)                               # x_0 = the start input (:202)
    dge = dg_e.view(B * L, 4 * H)
    hprev_e = torch.cat([torch.zeros_like(enc_out[:, :1]), enc_out[:, :-1]], 1).reshape(B * L, H).contiguous()
    g["encoder.weight_hh_l0"] = ops.gemm(dge, hprev_e, True, True)
    g["encoder.weight_ih_l0"] = ops.gemm(dge, embedded, True, True)
    g["encoder.bias_ih_l0"] = ops.colsum(dge)
    g["encoder.bias_hh_l0"] = g["encoder.bias_ih_l0"]
    d_emb = ops.gemm(dge, p["encoder.weight_ih_l0"], False, True).view(B, L, H)                        # dG . W_ih
    ops.scatter_dx(dx, idx, d_emb)                                                                     # + the decoder inputs (:235)
    d_emb2 = d_emb.view(B * L, H)
    g["embedding2.weight"] = ops.gemm(d_emb2, flat, True, True)
    g["embedding2.bias"] = ops.colsum(d_emb2)
    if E:                                           # d embedding1: the first E columns of d(embedding2's input), summed per category
        d_flat = ops.gemm(d_emb2, p["embedding2.weight"], False, True)                                 # [B L, H] . [H, E + 8]
        g["embedding1.weight"] = ops.embed_grad(d_flat, rows9, E, p["embedding1.weight"].shape[0])
    return g, d["logp"]


class ActorAdam:
    """torch.optim.Adam(model.actor.parameters(), lr) (trainPNHigh.py:62) + clip_grad_norm_ (:105-106), state on the device."""

    def __init__(self, actor, lr=0.5e-4, max_grad_norm=2.0):
        self.actor, self.lr, self.max_grad_norm, self.steps = actor, lr, max_grad_norm, 0
        self.state = {k: (torch.zeros_like(p.data, dtype=torch.float32), torch.zeros_like(p.data, dtype=torch.float32))
                      for k, p in _params(actor).items()}

    def state_dict(self):
        """torch.optim.Adam's layout (what the reference saves as actor_optim.state_dict(), trainPNHigh.py:118-123):
        parameters numbered in actor.parameters() order, state[i] = {step, exp_avg, exp_avg_sq}."""
        names = list(_params(self.actor))
        state = {i: {"step": self.steps, "exp_avg": self.state[k][0].detach().cpu().clone(),
                     "exp_avg_sq": self.state[k][1].detach().cpu().clone()} for i, k in enumerate(names)} if self.steps else {}
        group = {"lr": self.lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        names = list(_params(self.actor))
        group = sd["param_groups"][0]
        if list(group["params"]) != list(range(len(names))):
            raise ValueError("ActorAdam.load_state_dict: the optimizer state belongs to a different parameter list")
        self.lr = float(group["lr"])
        steps = {int(st["step"]) for st in sd["state"].values()}
        if len(steps) > 1:
            raise ValueError("ActorAdam.load_state_dict: parameters with different step counts")
        self.steps = steps.pop() if steps else 0
        for i, k in enumerate(names):
            if i in sd["state"]:
                m, v = self.state[k]
                m.copy_(sd["state"][i]["exp_avg"])
                v.copy_(sd["state"][i]["exp_avg_sq"])

    @torch.no_grad()
    def step(self, grads):
        """-> gradient norm before clipping (device tensor [1], float64)."""
        self.steps += 1
        extra = tuple(k for k in grads if k not in ACTOR_KEYS)         # embedding1.weight (embeddingTag=1), pointer.* / glimpse.* ('Bahdanau')
        uniq = [grads[k] for k in ACTOR_KEYS + extra]                   # b_ih and b_hh share a tensor but both count (two parameters)
        sumsq = ops.grad_sumsq(uniq)
        for k, p in _params(self.actor).items():
            if p.data.dtype != torch.float32 or not p.data.is_contiguous():
                raise ops.GnnpnError(f"ActorAdam: parameter {k} must be contiguous fp32")
            m, v = self.state[k]
            ops.adam_step(p.data, grads[k].contiguous(), m, v, sumsq, self.max_grad_norm, self.lr, self.steps)
        self.actor._packed = None                                  # the inference kernels re-pack the new weights
        self.actor.fold_on_device = True                           # ... forming the folded input matrices on the device
        return sumsq.sqrt()


class TrainModel:
    """TrainModel (trainPNHigh.py:43-150).  ``train_step`` is the body of the batch loop (:81-110)."""

    def __init__(self, model, train_dataset, val_dataset, epochDiv, beta, USE_CUDA, dataset, serCategory, lr=0.5e-4,
                 batch_size=128, threshold=None, max_grad_norm=2., low_model=None, device="cuda:0"):
        self.model, self.low_model = model, low_model
        self.train_dataset, self.val_dataset = train_dataset, val_dataset
        self.batch_size, self.threshold, self.epochDiv, self.beta = batch_size, threshold, epochDiv, beta
        self.USE_CUDA, self.dataset, self.serCategory = USE_CUDA, dataset, serCategory
        self.actor_optim = ActorAdam(model.actor, lr, float(max_grad_norm))
        self.max_grad_norm = float(max_grad_norm)
        self.train_tour, self.val_tour, self.epochs = [], [], 0
        self.device = torch.device(device)
        self.critic_exp_mvg_avg = None

    @torch.no_grad()
    def train_step(self, inputs, sample_seed=None):
        """One batch: -> dict(R [B], loss, grad_norm, idx_high, logp).  ``sample_seed`` fixes the draws (tests); by default
        they follow the High actor's own stream (modelPN.PointerNet.next_sample_seed)."""
        from .modelPN import two_level_greedy
        inputs = inputs.to(self.device).contiguous()
        ha = self.model.actor
        seed = ha.next_sample_seed() if sample_seed is None else int(sample_seed)
        if self.low_model is not None:
            out = two_level_greedy(self.low_model, self.model, inputs, sample_high_seed=seed)      # :83-84
        else:                                   # the PNLow trainer (trainPNLow.py:80-84): one net, no latent
            o = ha.run(inputs, None, sample_seed=seed)
            out = {"idx_high": o["idx"], "idx_low": o["idx"], "win_low": None,
                   "R": torch.ops.gnnpn.qos_reward(o["actions"], 0 if self.model.level == "Low" else 1)}
        R = out["R"]
        if self.critic_exp_mvg_avg is None:                                                        # :87-90
            self.critic_exp_mvg_avg = R.mean()
        else:
            self.critic_exp_mvg_avg = self.critic_exp_mvg_avg * self.beta + (1.0 - self.beta) * R.mean()
        advantage = R - self.critic_exp_mvg_avg                                                    # :92
        B = inputs.shape[0]
        # the gradient wrt log p is advantage/B (mean over the batch, :100-101); it needs the log-prob sums only for the
        # reference's "logprobs[logprobs < -1000] = 0" (:98), so the teacher-forced forward runs first with unit scale
        grads, logp = actor_gradients(ha, inputs, out["idx_high"], out["win_low"], (advantage / B).contiguous())
        logprobs = logp.sum(1)
        dead = logprobs < -1000
        if bool(dead.any()):                                                                       # rare: redo without those problems
            gscale = torch.where(dead, torch.zeros_like(advantage), advantage / B).contiguous()
            grads, logp = actor_gradients(ha, inputs, out["idx_high"], out["win_low"], gscale)
            logprobs = torch.where(dead, torch.zeros_like(logprobs), logprobs)
        loss = (advantage * logprobs).mean()
        norm = self.actor_optim.step(grads)                                                        # :103-108
        self.critic_exp_mvg_avg = self.critic_exp_mvg_avg.detach()
        self.train_tour.append(float(R.mean().item()))                                             # :112
        ops.check_status(self.device)
        return {"R": R, "loss": loss, "grad_norm": norm, "idx_high": out["idx_high"], "idx_low": out["idx_low"], "logp": logp,
                "grads": grads}

    def train_and_validate(self, n_epochs, epochDiv):
        """trainPNHigh.py:68-150 without the plotting: epochs over the training set, the evaluation block and the
        checkpoints / allActions artefacts of every ``epochDiv``-th epoch."""
        import json
        loader = torch.utils.data.DataLoader(self.train_dataset, batch_size=self.batch_size, shuffle=True, num_workers=0)
        if self.low_model is None:
            return self._train_and_validate_low(loader, n_epochs, epochDiv)
        t0 = time.time()                                                                            # :75
        for epoch in range(1, n_epochs + 1):
            for batch_id, (sample_batch, _labs) in enumerate(loader):
                if batch_id == 0:
                    self.critic_exp_mvg_avg = None                                                 # :87-88 restarts every epoch
                self.train_step(sample_batch)
            if self.threshold and self.train_tour[-1] < self.threshold:
                print("EARLY STOPPAGE!")
                break
            if epoch % epochDiv == 0:
                n = self.epochs // epochDiv
                os.makedirs(f"./solutions/PNHigh/{self.dataset}", exist_ok=True)
                torch.save({"epoch": epoch, "model": self.model.state_dict(), "optimizer": self.actor_optim.state_dict()},
                           f"./solutions/PNHigh/{self.dataset}/epoch{n}.model")                    # :118-123
                torch.save({"epoch": epoch, "model": self.low_model.state_dict(), "optimizer": self.actor_optim.state_dict()},
                           f"./solutions/PNHigh/{self.dataset}/epoch{n}_low.model")                # :124-129
                acts, tour = evalPN.evaluate(self.low_model, self.model, self.val_dataset, self.serCategory, 128,
                                             str(self.device))                                     # :131-141
                with open(f"./solutions/PNHigh/{self.dataset}/allActions{n}.txt", "w") as f:       # :143-144
                    json.dump(acts, f)
                self.val_tour += [float(r) for r in tour]                                          # :141, one mean per eval batch
                with open(f"./solutions/PNHigh/{self.dataset}/val{n}.txt", "w") as f:              # :147-148
                    json.dump(self.val_tour, f)
                with open(f"./solutions/PNHigh/{self.dataset}/time{n}.txt", "w") as f:             # :149-150
                    json.dump([time.time() - t0], f)
            self.epochs += 1

    def _train_and_validate_low(self, loader, n_epochs, epochDiv):
        """trainPNLow.py:71-147 (the Low net alone): checkpoints, allActions / allR / val artefacts of every epochDiv-th epoch."""
        import json
        for epoch in range(1, n_epochs + 1):
            for batch_id, (sample_batch, _labs) in enumerate(loader):
                if batch_id == 0:
                    self.critic_exp_mvg_avg = None
                self.train_step(sample_batch)
            if self.threshold and self.train_tour[-1] < self.threshold:
                print("EARLY STOPPAGE!")
                break
            if epoch % epochDiv == 0:
                n = self.epochs // epochDiv
                os.makedirs(f"./solutions/PNLow/{self.dataset}", exist_ok=True)
                torch.save({"epoch": epoch, "model": self.model.state_dict(), "optimizer": self.actor_optim.state_dict()},
                           f"./solutions/PNLow/{self.dataset}/epoch{n}.model")                     # trainPNLow.py:112-117
                all_actions = [[] for _ in range(self.serCategory + 2)]                            # :122
                all_r = {"quality": [], "averageQ": 0}
                for lo in range(0, len(self.val_dataset), 128):                                    # :127-135 (default sample="sample")
                    x = torch.stack([self.val_dataset[i][0] for i in range(lo, min(len(self.val_dataset), lo + 128))])
                    R, _, actions, _, _ = self.model(x.to(self.device), None)
                    all_r["quality"] += R.cpu().numpy().tolist()
                    for a in range(len(actions)):
                        all_actions[a] += actions[a].cpu().numpy().tolist()
                    self.val_tour.append(float(R.mean().item()))
                ops.check_status(self.device)
                with open(f"./solutions/PNLow/{self.dataset}/allActions{n}.txt", "w") as f:
                    json.dump(all_actions, f)
                with open(f"./solutions/PNLow/{self.dataset}/allR{n}.txt", "w") as f:
                    if all_r["quality"]:
                        all_r["averageQ"] = sum(all_r["quality"]) / len(all_r["quality"])
                        json.dump(all_r, f)
                with open(f"./solutions/PNLow/{self.dataset}/val{n}.txt", "w") as f:
                    json.dump(self.val_tour, f)
            self.epochs += 1


def _pointer_model(level, hidden_size, n_glimpses, tanh_exploration, use_tanh, serNumber, serCategory, use_cuda, embeddingTag=0):
    from .modelPN import CombinatorialRL, reward
    embedding_size = 20 if embeddingTag else 0                                                     # trainPNHigh.py:197-201, trainPNLow.py:196-199
    return CombinatorialRL(embedding_size, hidden_size, serCategory * serNumber, n_glimpses, tanh_exploration, use_tanh, reward,
                           attention="Dot", level=level, use_cuda=use_cuda, sNumber=serNumber, sCategory=serCategory)


def artefact_dataset(dataset, embeddingTag):
    """The dataset part of every artefact path of a PNLow / PNHigh run, WITH its trailing slash: the reference appends
    "20embeddings/" when the category embedding is on (trainPNHigh.py:197-198, trainPNLow.py:190-191: ``self.dataset += "20embeddings/"``
    after the candidate rows have been loaded from the plain dataset directory), so that an embeddingTag=1 run writes its checkpoints,
    allActions, val and time files under ./solutions/PN{Low,High}/<ds>/20embeddings/ and PNHigh finds its Low net there — beside, not
    over, the embeddingTag=0 artefacts.  (The reference mutates self.dataset, so a second start() would append twice; this is a
    function of the constructor's value.)"""
    return dataset + ("20embeddings/" if embeddingTag else "")


def low_checkpoint_path(ds, epochPNLow):
    """Where PNHigh loads its Low net from (trainPNHigh.py:237-240); ``ds`` = artefact_dataset(...).  With the embedding on the
    pretrained file is ./solutions/pretrained/<ds>/20embeddings-PNLow.model, as the reference's ``self.dataset[:-1]`` gives."""
    return (f"./solutions/PNLow/{ds}/epoch{epochPNLow}.model" if epochPNLow >= 0 else f"./solutions/pretrained/{ds[:-1]}-PNLow.model")


class PNHigh:
    """PNHigh (trainPNHigh.py:175-251): same constructor; ``start`` loads the candidate rows (loadDataPN), the trained Low
    net (./solutions/PNLow/<ds>/epoch{n}.model or ./solutions/pretrained/<ds>-PNLow.model) and trains the High net."""

    def __init__(self, dataset, embeddingTag, USE_CUDA, serCategory, epochDiv, serNumber, hidden_size, n_glimpses,
                 tanh_exploration, use_tanh, beta, max_grad_norm, lr, epochML, epochPNLow):
        self.embeddingTag = int(bool(embeddingTag))                                                # :179 (round 5: trained too)
        self.dataset = dataset + "/"
        self.USE_CUDA, self.serCategory, self.epochDiv, self.serNumber = USE_CUDA, serCategory, epochDiv, serNumber
        self.hidden_size, self.n_glimpses, self.tanh_exploration, self.use_tanh = hidden_size, n_glimpses, tanh_exploration, use_tanh
        self.beta, self.max_grad_norm, self.lr, self.epochML, self.epochPNLow = beta, max_grad_norm, lr, epochML, epochPNLow

    def start(self, n_epochs=100, device="cuda:0", batch_size=128):
        from .loadData import loadDataPN
        rows, labels = loadDataPN(epoch=self.epochML, dataset=self.dataset[:-1], serviceNumber=self.serNumber)   # :196
        ds = artefact_dataset(self.dataset, self.embeddingTag)                                     # :197-198
        n_train = len(rows) // 4 * 3
        train_ds = evalPN.SCDataset(rows[:n_train], labels[:n_train], bool(self.embeddingTag))     # :203-204
        val_ds = evalPN.SCDataset(rows[n_train:], labels[n_train:], bool(self.embeddingTag))
        args = (self.hidden_size, self.n_glimpses, self.tanh_exploration, self.use_tanh, self.serNumber, self.serCategory, self.USE_CUDA,
                self.embeddingTag)
        low, high = _pointer_model("Low", *args), _pointer_model("High", *args)
        root = low_checkpoint_path(ds, self.epochPNLow)                                            # :237-240
        low.load_state_dict(torch.load(root, map_location="cpu")["model"])                         # :241-242
        dev = torch.device(device)
        tm = TrainModel(high.to(dev), train_ds, val_ds, self.epochDiv, self.beta, self.USE_CUDA, ds[:-1], self.serCategory,
                        self.lr, batch_size, None, self.max_grad_norm, low_model=low.to(dev), device=device)
        tm.train_and_validate(n_epochs, self.epochDiv)                                             # :251
        return tm


class PNLow:
    """PNLow (trainPNLow.py:172-223): same constructor; trains the Low net (reward = number of violated constraints)."""

    def __init__(self, dataset, embeddingTag, USE_CUDA, serCategory, epochDiv, serNumber, hidden_size, n_glimpses,
                 tanh_exploration, use_tanh, beta, max_grad_norm, lr, epochML):
        self.embeddingTag = int(bool(embeddingTag))                                                # trainPNLow.py:173
        self.dataset = dataset + "/"
        self.USE_CUDA, self.serCategory, self.epochDiv, self.serNumber = USE_CUDA, serCategory, epochDiv, serNumber
        self.hidden_size, self.n_glimpses, self.tanh_exploration, self.use_tanh = hidden_size, n_glimpses, tanh_exploration, use_tanh
        self.beta, self.max_grad_norm, self.lr, self.epochML = beta, max_grad_norm, lr, epochML

    def start(self, n_epochs=50, device="cuda:0", batch_size=128):
        from .loadData import loadDataPN
        rows, labels = loadDataPN(epoch=self.epochML, dataset=self.dataset[:-1], serviceNumber=self.serNumber)   # :190-191
        ds = artefact_dataset(self.dataset, self.embeddingTag)                                     # trainPNLow.py:190-191
        n_train = len(rows) // 4 * 3
        train_ds = evalPN.SCDataset(rows[:n_train], labels[:n_train], bool(self.embeddingTag))     # trainPNLow.py:193-194
        val_ds = evalPN.SCDataset(rows[n_train:], labels[n_train:], bool(self.embeddingTag))
        model = _pointer_model("Low", self.hidden_size, self.n_glimpses, self.tanh_exploration, self.use_tanh, self.serNumber,
                               self.serCategory, self.USE_CUDA, self.embeddingTag)
        dev = torch.device(device)
        tm = TrainModel(model.to(dev), train_ds, val_ds, self.epochDiv, self.beta, self.USE_CUDA, ds[:-1], self.serCategory,
                        self.lr, batch_size, None, self.max_grad_norm, low_model=None, device=device)
        tm.train_and_validate(n_epochs, self.epochDiv)                                             # :223
        return tm
