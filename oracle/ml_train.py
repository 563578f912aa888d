"""Oracle (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py): ONE training step of the GNN candidate-ranking model, restated
on PyTorch-CPU autograd from /root/reference/src/models/trainML.py:34-47:

    self.model.train()                                              trainML.py:35
    x = self.model(data).squeeze()                                  :41   (Net.forward, modelML.py:131-176, BatchNorm on batch statistics)
    loss = BCELoss()(x, data.y.view(x.size(0), x.size(1)))          :42
    loss.backward() ; self.optimizer.step()                         :43,45  (Adam(lr), trainML.py:130)

The forward is oracle/ml.py's arithmetic (same conv restatements, "parity unpinned" at the torch_geometric boundary) with
BatchNorm1d in TRAINING mode: batch mean and biased variance normalise, the running statistics move by momentum 0.1 towards
the batch mean and the UNBIASED variance (torch.nn.BatchNorm1d defaults).  Parameters that take no part in the forward
(embedding tables 1..8 of the two NodeEncoders, noServicesLins) have no gradient and Adam leaves them alone.
Pinned by tests/golden/make_golden.py::gen_ml_train against the reference's own Net glue under autograd.
"""
import torch
import torch.nn.functional as F

from . import ml as oml

BN_MOMENTUM = 0.1


def trainable_keys(sd, n_gin, n_gcn, is_services=True):
    """The parameters that receive a gradient, in a fixed order (``is_services=False``: the graph-free ablation of
    modelML.py:157-162 — noServicesLins instead of the serviceConvs)."""
    keys = ["nodeEncoder.embeddings.0.weight", "serviceEncoder.embeddings.0.weight"]
    for i in range(n_gin):
        p = f"nodeConvs.{i}"
        keys += [f"{p}.eps", f"{p}.nn.0.weight", f"{p}.nn.0.bias", f"{p}.nn.1.weight", f"{p}.nn.1.bias", f"{p}.nn.3.weight",
                 f"{p}.nn.3.bias", f"nodeBatchNorms.{i}.weight", f"nodeBatchNorms.{i}.bias"]
    keys += ["nodeLin.weight", "nodeLin.bias"]
    for i in range(n_gcn):
        conv = f"serviceConvs.{i}" if is_services else f"noServicesLins.{i}"
        keys += [f"{conv}.weight", f"{conv}.bias", f"serviceBatchNorms.{i}.weight", f"serviceBatchNorms.{i}.bias"]
    keys += ["serviceLin.weight", "serviceLin.bias"]
    assert all(k in sd for k in keys)
    return keys


def bn_prefixes(n_gin, n_gcn):
    return [f"nodeConvs.{i}.nn.1" for i in range(n_gin)] + [f"nodeBatchNorms.{i}" for i in range(n_gin)] + \
           [f"serviceBatchNorms.{i}" for i in range(n_gcn)]


def _bn_train(x, p, prefix, stats):
    """BatchNorm1d, training mode: normalise with the batch mean / biased variance; record what the running buffers
    become (momentum 0.1, unbiased variance)."""
    mean = x.mean(0)
    var = x.var(0, unbiased=False)
    n = x.shape[0]
    stats[prefix] = (mean.detach(), (var * (n / max(n - 1, 1))).detach())
    return (x - mean) / torch.sqrt(var + oml.BN_EPS) * p[prefix + ".weight"] + p[prefix + ".bias"]


def train_forward(p, data, n_gin, n_gcn, S, stats=None, is_services=True):
    """Net.forward in training mode on a batched ``data`` (x_service = one copy per graph, edges as the batching left them)
    -> sigmoid scores [B,S].  ``p``: name -> tensor (requires_grad where wanted)."""
    stats = {} if stats is None else stats
    n_graphs = int(data.batch.max()) + 1
    ids = data.x[:, 0].long()
    h = torch.cat([p["nodeEncoder.embeddings.0.weight"][ids], data.x[:, 1:]], -1)                    # modelML.py:134-137
    for i in range(n_gin):                                                                            # :139-142
        pre = f"nodeConvs.{i}"
        agg = oml.scatter_sum(h[data.edge_index[0]], data.edge_index[1], h.shape[0])
        out = agg + (1 + p[pre + ".eps"]) * h
        out = F.linear(out, p[pre + ".nn.0.weight"], p[pre + ".nn.0.bias"])
        out = F.relu(_bn_train(out, p, pre + ".nn.1", stats))
        out = F.linear(out, p[pre + ".nn.3.weight"], p[pre + ".nn.3.bias"])
        h = F.relu(_bn_train(out, p, f"nodeBatchNorms.{i}", stats))
    h = F.linear(h, p["nodeLin.weight"], p["nodeLin.bias"])                                          # :165
    xr = oml.scatter_mean(h, data.batch, n_graphs)                                                    # :166
    sid = data.x_service[:, 0].long()
    xs = torch.cat([p["serviceEncoder.embeddings.0.weight"][sid], data.x_service[:, 1:]], -1)        # :146-149
    for i in range(n_gcn):                                                                            # :152-155 | :157-162
        if is_services:
            xs = oml.gcn_conv(xs, data.edge_index_service, data.edge_attr_service, p[f"serviceConvs.{i}.weight"],
                              p[f"serviceConvs.{i}.bias"])
        else:
            xs = F.linear(xs, p[f"noServicesLins.{i}.weight"], p[f"noServicesLins.{i}.bias"])
        xs = F.relu(_bn_train(xs, p, f"serviceBatchNorms.{i}", stats))
    xs = F.linear(xs, p["serviceLin.weight"], p["serviceLin.bias"])                                  # :164
    xs = oml.scatter_mean(xs, torch.arange(S).repeat(n_graphs), S)                                    # :167-172
    return torch.sigmoid(torch.matmul(xr, xs.t()))                                                    # :173-176


def train_step(sd, data, y, n_gin, n_gcn, S, lr, adam_state=None, step=1, is_services=True):
    """trainML.py:39-45 for ONE batch.  Returns dict(loss, scores, grads {name: tensor}, new_params {name: tensor},
    running {bn prefix: (running_mean, running_var)}, adam_state)."""
    keys = trainable_keys(sd, n_gin, n_gcn, is_services)
    p = {k: v.clone() for k, v in sd.items()}
    for k in keys:
        p[k].requires_grad_(True)
    stats = {}
    scores = train_forward(p, data, n_gin, n_gcn, S, stats, is_services)
    loss = F.binary_cross_entropy(scores, y.view(scores.shape))                                      # :42
    grads = dict(zip(keys, torch.autograd.grad(loss, [p[k] for k in keys])))
    st = adam_state or {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in keys}
    b1, b2, eps = 0.9, 0.999, 1e-8                                                                   # Adam defaults (:130)
    new_params, new_state = {}, {}
    for k in keys:
        g = grads[k]
        m = st[k][0] * b1 + (1 - b1) * g
        v = st[k][1] * b2 + (1 - b2) * g * g
        new_params[k] = sd[k] - lr * (m / (1 - b1 ** step)) / ((v / (1 - b2 ** step)).sqrt() + eps)
        new_state[k] = (m, v)
    running = {}
    for pre, (mean, var_u) in stats.items():
        running[pre] = ((1 - BN_MOMENTUM) * sd[pre + ".running_mean"] + BN_MOMENTUM * mean,
                        (1 - BN_MOMENTUM) * sd[pre + ".running_var"] + BN_MOMENTUM * var_u)
    return {"loss": loss.detach(), "scores": scores.detach(), "grads": grads, "new_params": new_params, "running": running,
            "adam_state": new_state}
