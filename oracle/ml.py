"""Oracle (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py): the GNN candidate-ranking
model, restated on PyTorch-CPU fp32 from /root/reference/src/models/modelML.py and
/root/reference/src/models/trainML.py:49-72.

Pinning status: the reference's ``Net`` glue (embedding, concat, BN, ReLU, pooling, matmul,
sigmoid; modelML.py:131-176) is run unmodified by tests/golden/make_golden.py and agrees with
``net_forward`` here.  ``GINConv``/``GCNConv`` (torch_geometric==1.7.0) and ``scatter``
(torch_scatter==2.0.6) — requirements.txt:6-7, call sites modelML.py:91,100,103,140,153,166,172 —
are absent from /root/reference and from this image; their arithmetic below restates the
published algorithm  => **parity unpinned** at that boundary.

Service-branch semantics, two forms:
* ``net_forward``: the service graph / service embedding is problem independent and is evaluated ONCE (what the
  device pipeline does; the mathematically intended model);
* ``net_forward_batched``: the reference's forward LITERALLY on whatever a batched ``data`` object holds — the GCN runs
  over all ``x_service`` rows (B concatenated copies of the table, trainML.py:109-114) with ``edge_index_service`` as
  the batching produced it, then ``scatter(..., serviceBatch, reduce='mean')`` averages row s of every copy
  (modelML.py:145-156,164,167-172).  ``pyg_batch_service_edges`` builds the edge list the way torch_geometric 1.7.0's
  ``Batch.from_data_list`` would: ``Data.__inc__`` offsets every key containing "index" by ``data.num_nodes`` — the
  WORKFLOW node count x.size(0), not S — so copy b's service edges are shifted by the number of workflow nodes of the
  graphs before it and land (mostly) among copy 0's rows.  [From memory of the 1.7.0 source; the package is not in
  this image, so this too is "parity unpinned".]
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def make_state_dict(hidden=128, emb=20, n_gin=2, n_gcn=2, seed=0, vocab=100, randomize_bn=True):
    """Deterministic weights in the reference's ``Net.state_dict()`` layout (modelML.py:56-115).
    BN running stats / affine are randomised (a freshly constructed BN is the identity and would
    hide epilogue bugs).  ``vocab`` > 100 gives the enlarged embedding tables the synthetic
    1000/2000-task configs need (Embedding(100, c) at modelML.py:16 cannot index them)."""
    g = torch.Generator().manual_seed(seed)

    def u(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def lin(prefix, out_f, in_f, sd):
        k = 1.0 / math.sqrt(in_f)
        sd[prefix + ".weight"] = u((out_f, in_f), k)
        sd[prefix + ".bias"] = u((out_f,), k)

    def bn(prefix, ch, sd):
        if randomize_bn:
            sd[prefix + ".weight"] = 1.0 + u((ch,), 0.3)
            sd[prefix + ".bias"] = u((ch,), 0.2)
            sd[prefix + ".running_mean"] = u((ch,), 0.2)
            sd[prefix + ".running_var"] = 1.0 + u((ch,), 0.5)
        else:
            sd[prefix + ".weight"] = torch.ones(ch)
            sd[prefix + ".bias"] = torch.zeros(ch)
            sd[prefix + ".running_mean"] = torch.zeros(ch)
            sd[prefix + ".running_var"] = torch.ones(ch)
        sd[prefix + ".num_batches_tracked"] = torch.tensor(0)

    sd = {}
    for enc in ("nodeEncoder", "serviceEncoder"):
        for i in range(9):                                               # modelML.py:15-16
            sd[f"{enc}.embeddings.{i}.weight"] = torch.randn((vocab, emb), generator=g)
    for i in range(n_gin):                                               # modelML.py:75-92
        in_f = emb + 6 if i == 0 else hidden
        sd[f"nodeConvs.{i}.eps"] = u((1,), 0.1)
        lin(f"nodeConvs.{i}.nn.0", 2 * hidden, in_f, sd)
        bn(f"nodeConvs.{i}.nn.1", 2 * hidden, sd)
        lin(f"nodeConvs.{i}.nn.3", hidden, 2 * hidden, sd)
        bn(f"nodeBatchNorms.{i}", hidden, sd)
    lin("nodeLin", hidden, hidden, sd)                                   # :93
    for i in range(n_gcn):                                               # :98-104
        in_f = emb + 4 if i == 0 else 2 * hidden
        bound = math.sqrt(6.0 / (in_f + 2 * hidden))                     # glorot
        sd[f"serviceConvs.{i}.weight"] = u((in_f, 2 * hidden), bound)    # stored in x out (PyG 1.7)
        sd[f"serviceConvs.{i}.bias"] = u((2 * hidden,), 0.1)
        bn(f"serviceBatchNorms.{i}", 2 * hidden, sd)
        lin(f"noServicesLins.{i}", 2 * hidden, in_f, sd)                 # :108-115 (the isServices=False branch)
    lin("serviceLin", hidden, 2 * hidden, sd)                            # :106
    return sd


def _bn(x, sd, prefix):
    """BatchNorm1d in eval mode (modelML.py:141,154 and inside the GIN MLP :79,87)."""
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, BN_EPS)


def scatter_sum(src, index, n):
    """torch_scatter.scatter(reduce='sum') along dim 0: CPU scatter_add_ visits rows in order, so
    each destination accumulates its contributions in edge order."""
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.index_add_(0, index, src)


def scatter_mean(src, index, n):
    """torch_scatter.scatter(reduce='mean') (call sites modelML.py:166,172): sum / clamp(count,1)."""
    total = scatter_sum(src, index, n)
    count = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(index.numel(), dtype=src.dtype))
    return total / count.clamp(min=1).unsqueeze(1)


def gin_conv(x, edge_index, eps, sd, prefix):
    """GINConv(nn, train_eps=True) (PyG 1.7.0; call site modelML.py:91,140):
    nn( scatter_add_{j->i}(x_j) + (1+eps) * x_i ), messages flow edge_index[0] -> edge_index[1]."""
    agg = scatter_sum(x[edge_index[0]], edge_index[1], x.shape[0])
    out = agg + (1 + eps) * x
    out = F.linear(out, sd[prefix + ".nn.0.weight"], sd[prefix + ".nn.0.bias"])
    out = F.relu(_bn(out, sd, prefix + ".nn.1"))
    return F.linear(out, sd[prefix + ".nn.3.weight"], sd[prefix + ".nn.3.bias"])


def gcn_norm(edge_index, edge_weight, n):
    """gcn_norm with add_remaining_self_loops(fill=1) (PyG 1.7.0): existing self loops keep their
    weight, every other node gets a weight-1 loop, loops are appended AFTER the non-loop edges;
    deg = scatter_add(w, col); norm = deg^-1/2[row] * w * deg^-1/2[col] with inf -> 0."""
    row, col = edge_index[0], edge_index[1]
    keep = row != col
    loop_w = torch.ones(n, dtype=edge_weight.dtype)
    if (~keep).any():
        loop_w[row[~keep]] = edge_weight[~keep]
    loops = torch.arange(n, dtype=row.dtype)
    row2 = torch.cat([row[keep], loops])
    col2 = torch.cat([col[keep], loops])
    w2 = torch.cat([edge_weight[keep], loop_w])
    deg = torch.zeros(n, dtype=w2.dtype).index_add_(0, col2, w2)
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    return row2, col2, dis[row2] * w2 * dis[col2]


def gcn_conv(x, edge_index, edge_weight, weight, bias):
    """GCNConv (PyG 1.7.0; call sites modelML.py:100,103,153): transform THEN aggregate:
    out_i = sum_e norm_e * (x @ W)[row_e]  over edges with col_e == i, in edge order; + bias."""
    n = x.shape[0]
    row, col, norm = gcn_norm(edge_index, edge_weight, n)
    xw = torch.matmul(x, weight)
    out = scatter_sum(norm.view(-1, 1) * xw[row], col, n)
    return out + bias


@torch.no_grad()
def service_embedding(sd, x_service, edge_index_service, edge_attr_service, n_gcn, is_services=True):
    """The service branch of Net.forward (modelML.py:145-164): [S,5] -> [S,hidden].  ``is_services`` False is the
    graph-free ablation (:157-162): a Linear per layer in place of the GCNConv, same batch norms."""
    ids = x_service[:, 0].long()
    xs = torch.cat([sd["serviceEncoder.embeddings.0.weight"][ids], x_service[:, 1:]], -1)  # :146-149
    for i in range(n_gcn):
        if is_services:                                                                     # :152-155
            xs = gcn_conv(xs, edge_index_service, edge_attr_service,
                          sd[f"serviceConvs.{i}.weight"], sd[f"serviceConvs.{i}.bias"])
        else:                                                                               # :158-159
            xs = F.linear(xs, sd[f"noServicesLins.{i}.weight"], sd[f"noServicesLins.{i}.bias"])
        xs = F.relu(_bn(xs, sd, f"serviceBatchNorms.{i}"))
    return F.linear(xs, sd["serviceLin.weight"], sd["serviceLin.bias"])                     # :164


@torch.no_grad()
def request_embedding(sd, x, edge_index, batch, n_graphs, n_gin):
    """The workflow branch of Net.forward (modelML.py:133-143,165-166): -> [B,hidden]."""
    ids = x[:, 0].long()
    h = torch.cat([sd["nodeEncoder.embeddings.0.weight"][ids], x[:, 1:]], -1)               # :134-137
    for i in range(n_gin):                                                                  # :139-142
        h = gin_conv(h, edge_index, sd[f"nodeConvs.{i}.eps"], sd, f"nodeConvs.{i}")
        h = F.relu(_bn(h, sd, f"nodeBatchNorms.{i}"))
    h = F.linear(h, sd["nodeLin.weight"], sd["nodeLin.bias"])                               # :165
    return scatter_mean(h, batch, n_graphs)                                                 # :166


@torch.no_grad()
def net_forward(sd, data, n_gin, n_gcn, is_services=True):
    """Net.forward (modelML.py:131-176) -> sigmoid scores [B,S]."""
    n_graphs = int(data.batch.max()) + 1
    xr = request_embedding(sd, data.x, data.edge_index, data.batch, n_graphs, n_gin)
    xs = service_embedding(sd, data.x_service, data.edge_index_service, data.edge_attr_service, n_gcn, is_services)
    return torch.sigmoid(torch.matmul(xr, xs.t()))                                          # :173-176


def pyg_batch_service_edges(edge_index_service, edge_attr_service, offsets):
    """The batched ``edge_index_service`` / ``edge_attr_service`` of B graphs that each carry the same service graph
    (trainML.py:109-114): copy b's edges shifted by ``offsets[b]``.
    torch_geometric 1.7.0 (``Data.__inc__`` -> ``num_nodes`` = ``x.size(0)``, see the module docstring): offsets =
    cumulative WORKFLOW node counts [0, N_0, N_0+N_1, ...];  B clean block-diagonal replicas: offsets = [0, S, 2S, ...]."""
    return (torch.cat([edge_index_service + int(o) for o in offsets], 1), edge_attr_service.repeat(len(offsets)))


@torch.no_grad()
def net_forward_batched(sd, data, n_gin, n_gcn, S):
    """Net.forward (modelML.py:131-176) literally on a batched ``data``: x_service [B*S,5] (B copies), edge_index_service /
    edge_attr_service as the batching left them -> sigmoid scores [B,S]."""
    n_graphs = int(data.batch.max()) + 1
    xr = request_embedding(sd, data.x, data.edge_index, data.batch, n_graphs, n_gin)
    xs = service_embedding(sd, data.x_service, data.edge_index_service, data.edge_attr_service, n_gcn)   # :145-156,164
    service_batch = torch.arange(S).repeat(n_graphs)                                                     # :167-171
    xs = scatter_mean(xs, service_batch, S)                                                              # :172
    return torch.sigmoid(torch.matmul(xr, xs.t()))                                                       # :173-176


def rank_services(scores):
    """Per-row descending ranking (trainML.py:62).  The reference's sort is unstable, i.e. tie
    order is undefined there; the oracle (and the build) define it as lowest index first."""
    return torch.sort(scores, dim=1, descending=True, stable=True).indices


def precision_at(ranking, labels, ks=(1, 5)):
    """P@k of trainML.py:63-70: fraction of the top-k ranked services whose label is 1."""
    out = []
    for k in ks:
        hit = torch.gather(labels, 1, ranking[:, :k]) == 1
        out.append(hit.float().sum(1) / k)
    return [float(o.mean()) for o in out]


def make_data(x, edge_index, batch, x_service, edge_index_service, edge_attr_service):
    return SimpleNamespace(x=x, edge_index=edge_index, batch=batch, x_service=x_service,
                           edge_index_service=edge_index_service, edge_attr_service=edge_attr_service)
