"""Training of the GNN candidate-ranking model behind the reference's entry point
(/root/reference/src/models/trainML.py): ``TrainML(dataset1, numLayersGIN, numLayersGCN, hiddenChannels,
embeddingChannels, dropout, lr, epochs)`` with ``train()`` (:34-47), ``test(loader)`` (:49-72) and ``start()`` (:74-149) —
SURVEY.md §8f row 4.

No autograd: ``ml_train_step`` is one batch of ``train()`` — the forward of ``Net`` in training mode (BatchNorm1d on batch
statistics, modelML.py:131-176), ``BCELoss`` (:42), the backward written out layer by layer, and ``Adam(lr)`` (:130) —
entirely on the kernels of libgnnpn_hip.so: the inference kernels (embed_concat, linear, csr_aggregate on the graph and on its
TRANSPOSE, segment_mean), the generic training kernels (gemm with k-major operands for the weight gradients and the
input gradients, colsum, adam_step) and csrc/train_ml.hip (BatchNorm forward/backward on batch statistics, BCE-of-sigmoid
loss + gradient, embedding-table gradient, dot for GINConv's eps).  torch supplies device memory and the layout sorts.

Batches are assembled as torch_geometric 1.7.0's ``DataLoader(batch_size=2)`` assembles them (trainML.py:121-122): node
features concatenated, workflow edges shifted by the node counts, one copy of the service table per graph, and the service
edges of graph b shifted by the WORKFLOW node count of the graphs before it (``Data.__inc__`` keys on "index" — the
reference's own batching, reproduced, not repaired).
"""
import json
import os
import time

import torch

from . import graph, ops
from .loadData import loadData
from .modelML import BN_EPS, Net
from .ops import ACT_SIGMOID

F32 = torch.float32
BN_MOMENTUM = 0.1


class MLBatch:
    """One DataLoader batch on the device, with every graph layout the step needs (forward AND transposed)."""

    def __init__(self, graphs, service, device):
        """graphs: list of dicts x [n,7] float, edge_index [2,e] long, y [S] float;  service: dict x_service [S,5],
        edge_index_service [2,E], edge_attr_service [E] (host tensors)."""
        B = len(graphs)
        S = service["x_service"].shape[0]
        counts = [g["x"].shape[0] for g in graphs]
        offs = [0]
        for c in counts[:-1]:
            offs.append(offs[-1] + c)
        self.B, self.S, self.N = B, S, sum(counts)
        self.x = torch.cat([g["x"].float() for g in graphs]).contiguous().to(device)
        ei = torch.cat([g["edge_index"].long() + o for g, o in zip(graphs, offs)], 1).to(device)
        batch = torch.cat([torch.full((c,), b, dtype=torch.long) for b, c in enumerate(counts)]).to(device)
        self.y = torch.cat([g["y"].float().reshape(-1) for g in graphs]).view(B, S).contiguous().to(device)
        self.wf = graph.csr_by_destination(ei, self.N)
        self.wf_t = graph.csr_by_destination(ei.flip(0), self.N)
        self.seg = graph.segment_ptr(batch, B)
        inv_count = (1.0 / torch.tensor(counts, dtype=F32, device=device))[batch].contiguous()
        ar = torch.arange(self.N + 1, dtype=torch.int32, device=device)
        self.pool_t = graph.CSR(ar, batch.to(torch.int32).contiguous(), inv_count, self.N)      # d(segment mean): row n <- graph of n
        # service side: B copies, copy b's edges shifted by the workflow node count before it (PyG 1.7.0 Data.__inc__)
        self.n_svc = B * S
        self.xs = service["x_service"].float().repeat(B, 1).contiguous().to(device)
        eis = torch.cat([service["edge_index_service"].long() + o for o in offs], 1).to(device)
        eas = service["edge_attr_service"].float().repeat(B).to(device)
        self.svc = graph.gcn_csr(eis, eas, self.n_svc)
        dst = torch.repeat_interleave(torch.arange(self.n_svc, device=device), (self.svc.rowptr[1:] - self.svc.rowptr[:-1]).long())
        order_src = torch.stack([dst, self.svc.col.long()])                                       # (dst -> src) = the transpose
        self._svc_t_perm = torch.sort(self.svc.col.long(), stable=True).indices
        self.svc_t = graph.csr_by_destination(order_src, self.n_svc)
        self.mean_rowptr = (torch.arange(S + 1, device=device, dtype=torch.int32) * B).contiguous()
        self.mean_col = (torch.arange(S, device=device, dtype=torch.int32).view(S, 1) +
                         S * torch.arange(B, device=device, dtype=torch.int32).view(1, B)).reshape(-1).contiguous()
        self.spread = graph.CSR(torch.arange(self.n_svc + 1, dtype=torch.int32, device=device),
                                (torch.arange(self.n_svc, device=device, dtype=torch.int32) % S).contiguous(),
                                torch.full((self.n_svc,), 1.0 / B, dtype=F32, device=device), self.n_svc)

    def norm_t(self, norm):
        """The GCN edge weights in the transposed CSR's order."""
        return norm[self._svc_t_perm].contiguous()


def _p(t):
    d = t.data
    if d.dtype != F32 or not d.is_contiguous():
        raise ops.GnnpnError("training needs contiguous fp32 parameters")
    return d


def _bn_fwd(x, bn, relu=True):
    y, xhat, invstd = ops.bn_train_forward(x, _p(bn.weight), _p(bn.bias), relu, bn.running_mean, bn.running_var, BN_EPS, BN_MOMENTUM)
    bn.num_batches_tracked += 1
    return y, xhat, invstd


def ml_forward_backward(net, b):
    """Forward in training mode, BCE loss, backward.  -> (loss [1] device tensor, scores [B,S], {parameter name: gradient})."""
    c = net.reqAndServiceChannels
    g = {}
    # ---------------- forward: workflow branch (modelML.py:133-143,165-166)
    node_table = _p(net.nodeEncoder.embeddings[0].weight)
    h = ops.embed_concat(b.x, node_table)
    gin = []
    for conv, bn in zip(net.nodeConvs, net.nodeBatchNorms):
        a = ops.csr_aggregate(b.wf.rowptr, b.wf.col, None, h, self_coef=_p(conv.eps))
        u = ops.linear(a, _p(conv.nn[0].weight), _p(conv.nn[0].bias))
        r, xh1, is1 = _bn_fwd(u, conv.nn[1])
        w = ops.linear(r, _p(conv.nn[3].weight), _p(conv.nn[3].bias))
        t, xh2, is2 = _bn_fwd(w, bn)
        gin.append((h, a, r, xh1, is1, t, xh2, is2))
        h = t
    z = ops.linear(h, _p(net.nodeLin.weight), _p(net.nodeLin.bias))
    xr = ops.segment_mean(b.seg, z)
    # ---------------- forward: service branch (:145-156,164,167-172)
    s = ops.embed_concat(b.xs, _p(net.serviceEncoder.embeddings[0].weight))
    norm = ops.gcn_norm(b.svc.rowptr, b.svc.col, b.svc.w) if net.isService else None
    gcn = []
    for i, bn in enumerate(net.serviceBatchNorms):
        if net.isService:
            conv = net.serviceConvs[i]
            xw = ops.linear(s, _p(conv.weight).t().contiguous())                   # transform first (GCNConv)
            agg = ops.csr_aggregate(b.svc.rowptr, b.svc.col, norm, xw, bias=_p(conv.bias))
        else:                                                                      # the graph-free ablation (modelML.py:157-162)
            lin = net.noServicesLins[i]
            agg = ops.linear(s, _p(lin.weight), _p(lin.bias))
        q, xh, is_ = _bn_fwd(agg, bn)
        gcn.append((s, q, xh, is_))
        s = q
    e = ops.linear(s, _p(net.serviceLin.weight), _p(net.serviceLin.bias))
    emb = ops.csr_aggregate(b.mean_rowptr, b.mean_col, None, e) / float(b.B)       # scatter mean over the copies (:172)
    scores = ops.linear(xr, emb, act=ACT_SIGMOID)                                   # :173-176
    loss, dz = ops.bce_sigmoid(scores, b.y)                                         # trainML.py:42
    # ---------------- backward: head
    dxr = ops.gemm(dz, emb, b_kmajor=True)                                          # [B,h]
    demb = ops.gemm(dz, xr, a_kmajor=True, b_kmajor=True)                           # [S,h]
    de = ops.csr_aggregate(b.spread.rowptr, b.spread.col, b.spread.w, demb)         # every copy's row gets demb / B
    # ---------------- backward: service branch
    g["serviceLin.weight"] = ops.gemm(de, s, a_kmajor=True, b_kmajor=True)
    g["serviceLin.bias"] = ops.colsum(de)
    ds = ops.gemm(de, _p(net.serviceLin.weight), b_kmajor=True)
    norm_t = b.norm_t(norm) if net.isService else None
    for i in reversed(range(len(gcn))):
        s_in, q, xh, is_ = gcn[i]
        bn = net.serviceBatchNorms[i]
        dagg, g[f"serviceBatchNorms.{i}.weight"], g[f"serviceBatchNorms.{i}.bias"] = \
            ops.bn_train_backward(ds, q, xh, _p(bn.weight), is_, True)
        if not net.isService:                                                       # Linear: dW = dagg^T s_in, db = colsum, ds = dagg W
            lin = net.noServicesLins[i]
            g[f"noServicesLins.{i}.weight"] = ops.gemm(dagg, s_in, a_kmajor=True, b_kmajor=True)
            g[f"noServicesLins.{i}.bias"] = ops.colsum(dagg)
            ds = ops.gemm(dagg, _p(lin.weight), b_kmajor=True)
            continue
        conv = net.serviceConvs[i]
        g[f"serviceConvs.{i}.bias"] = ops.colsum(dagg)
        dxw = ops.csr_aggregate(b.svc_t.rowptr, b.svc_t.col, norm_t, dagg)          # the aggregate's transpose
        g[f"serviceConvs.{i}.weight"] = ops.gemm(s_in, dxw, a_kmajor=True, b_kmajor=True)   # [in,out] as stored
        ds = ops.gemm(dxw, _p(conv.weight))                                         # dxw . W^T
    g["serviceEncoder.embeddings.0.weight"] = ops.embed_grad(ds, b.xs, c, net.serviceEncoder.embeddings[0].weight.shape[0])
    # ---------------- backward: workflow branch
    dzn = ops.csr_aggregate(b.pool_t.rowptr, b.pool_t.col, b.pool_t.w, dxr)         # d(segment mean)
    g["nodeLin.weight"] = ops.gemm(dzn, h, a_kmajor=True, b_kmajor=True)
    g["nodeLin.bias"] = ops.colsum(dzn)
    dh = ops.gemm(dzn, _p(net.nodeLin.weight), b_kmajor=True)
    for i in reversed(range(len(gin))):
        h_in, a, r, xh1, is1, t, xh2, is2 = gin[i]
        conv, bn = net.nodeConvs[i], net.nodeBatchNorms[i]
        pre = f"nodeConvs.{i}"
        dw, g[f"nodeBatchNorms.{i}.weight"], g[f"nodeBatchNorms.{i}.bias"] = ops.bn_train_backward(dh, t, xh2, _p(bn.weight), is2, True)
        g[f"{pre}.nn.3.weight"] = ops.gemm(dw, r, a_kmajor=True, b_kmajor=True)
        g[f"{pre}.nn.3.bias"] = ops.colsum(dw)
        dr = ops.gemm(dw, _p(conv.nn[3].weight), b_kmajor=True)
        du, g[f"{pre}.nn.1.weight"], g[f"{pre}.nn.1.bias"] = ops.bn_train_backward(dr, r, xh1, _p(conv.nn[1].weight), is1, True)
        g[f"{pre}.nn.0.weight"] = ops.gemm(du, a, a_kmajor=True, b_kmajor=True)
        g[f"{pre}.nn.0.bias"] = ops.colsum(du)
        da = ops.gemm(du, _p(conv.nn[0].weight), b_kmajor=True)
        g[f"{pre}.eps"] = ops.dot(da, h_in)                                         # out = agg + (1 + eps) x
        dh = ops.csr_aggregate(b.wf_t.rowptr, b.wf_t.col, None, da, self_coef=_p(conv.eps))
    g["nodeEncoder.embeddings.0.weight"] = ops.embed_grad(dh, b.x, c, node_table.shape[0])
    return loss, scores, g


class MLAdam:
    """torch.optim.Adam(model.parameters(), lr) (trainML.py:130) on gnnpn_adam_step_f32; parameters without a gradient
    (embedding tables 1..8, noServicesLins) are left alone, as torch's Adam leaves them."""

    def __init__(self, net, lr):
        self.named = dict(net.named_parameters())
        self.lr, self.step_no = float(lr), 0
        self.state = {}
        self._zero = None

    def step(self, grads):
        self.step_no += 1
        for name, gr in grads.items():
            p = _p(self.named[name])
            if name not in self.state:
                self.state[name] = (torch.zeros_like(p), torch.zeros_like(p))
            if self._zero is None:
                self._zero = torch.zeros(1, dtype=torch.float64, device=p.device)      # squared norm 0: no clipping
            m, v = self.state[name]
            ops.adam_step(p, gr.reshape(p.shape).contiguous(), m, v, self._zero, 1.0, self.lr, self.step_no)


def ml_train_step(net, batch, adam):
    """One batch of TrainML.train (trainML.py:39-45); returns the loss as a device tensor [1]."""
    if not net.training:
        raise ops.GnnpnError("ml_train_step: call net.train() first (trainML.py:35)")
    loss, _, grads = ml_forward_backward(net, batch)
    adam.step(grads)
    net._prep = None                                                                # the inference constants are stale
    return loss


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor, patience, min_lr) with torch's defaults
    (threshold 1e-4 'rel', cooldown 0, eps 1e-8), as trainML.py:131-132 builds it — including its use of mode='min' on a
    precision (trainML.py:138: a quirk of the reference, kept)."""

    def __init__(self, adam, factor=0.5, patience=3, min_lr=0.00001, threshold=1e-4, eps=1e-8):
        self.adam, self.factor, self.patience, self.min_lr, self.threshold, self.eps = adam, factor, patience, min_lr, threshold, eps
        self.best, self.bad = float("inf"), 0

    def step(self, metric):
        metric = float(metric)
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            new = max(self.adam.lr * self.factor, self.min_lr)
            if self.adam.lr - new > self.eps:
                self.adam.lr = new
            self.bad = 0


class TrainML:
    """TrainML (trainML.py:15-149): same constructor, same members, same artefacts."""

    def __init__(self, dataset1, numLayersGIN, numLayersGCN, hiddenChannels, embeddingChannels, dropout, lr, epochs):
        self.dataset1 = dataset1
        self.hiddenChannels, self.embeddingChannels = hiddenChannels, embeddingChannels
        self.numLayersGIN, self.numLayersGCN = numLayersGIN, numLayersGCN
        self.epochs, self.dropout, self.lr = epochs, dropout, lr
        self.device = torch.device("cuda")
        self.train_loader = self.val_loader = None
        self.model = self.optimizer = None
        self.batch_size = 2                                                         # :121-122

    # a "loader" is a list of index lists into self.graphs; the train loader is reshuffled every epoch (shuffle=True)
    def _batches(self, idx, shuffle):
        idx = list(idx)
        if shuffle:
            perm = torch.randperm(len(idx)).tolist()
            idx = [idx[i] for i in perm]
        return [idx[i:i + self.batch_size] for i in range(0, len(idx), self.batch_size)]

    def _batch(self, ids):
        return MLBatch([self.graphs[i] for i in ids], self.service, self.device)

    def train(self):                                                                # :34-47
        self.model.train()
        total, n = 0.0, 0
        for ids in self._batches(self.train_idx, True):
            loss = ml_train_step(self.model, self._batch(ids), self.optimizer)
            total += float(loss.item()) * len(ids)                                  # loss.item() * data.num_graphs
            n += len(ids)
        return total / max(n, 1)

    @torch.no_grad()
    def test(self, idx):                                                            # :49-72
        self.model.eval()
        idx_list, pats = [], []
        for ids in self._batches(idx, False):
            b = self._batch(ids)
            data = _as_data(b)
            x = self.model(data)
            rank = ops.rank_rows(x)
            pats.append(ops.precision_at_k(rank, b.y, (1, 5)))
            idx_list += rank.cpu().tolist()
        pat = torch.cat(pats).mean(0).tolist() if pats else [0.0, 0.0]
        return idx_list, pat

    def start(self, out_dir=None):                                                  # :74-149
        nodefeatures, services, edge_indices, eis, eas, labels, _ = loadData(self.dataset1)
        self.graphs = [{"x": torch.tensor(nf, dtype=F32), "edge_index": torch.tensor(ei, dtype=torch.long).view(2, -1),
                        "y": torch.tensor(lab, dtype=F32)} for nf, ei, lab in zip(nodefeatures, edge_indices, labels)]
        self.service = {"x_service": torch.tensor(services, dtype=F32),
                        "edge_index_service": torch.tensor(eis, dtype=torch.long).view(2, -1),
                        "edge_attr_service": torch.tensor(eas, dtype=F32)}
        n = len(self.graphs)
        self.train_idx, self.val_idx = list(range(n // 4 * 3)), list(range(n // 4 * 3, n))    # :121-122
        t0 = time.time()
        self.model = Net(self.hiddenChannels, len(labels[0]), self.embeddingChannels, self.numLayersGIN, self.numLayersGCN,
                         isServices=True, dropout=0.0).to(self.device)              # :125-126
        print()
        print(f"Run {0}:")
        print()
        self.model.reset_parameters()                                               # :132
        self.optimizer = MLAdam(self.model, self.lr)                                # :130
        scheduler = ReduceLROnPlateau(self.optimizer, factor=0.5, patience=3, min_lr=0.00001)
        out_dir = out_dir or f"solutions/ML/{self.dataset1}/"
        os.makedirs(out_dir, exist_ok=True)
        for epoch in range(self.epochs):                                            # :134-149
            lr = self.optimizer.lr
            loss = self.train()
            val_idx_list, val_mae = self.test(self.val_idx)
            scheduler.step(val_mae[0])
            print(f"Epoch: {epoch:03d}, LR: {lr:.5f}, Loss: {loss:.4f}, ValP@1: {val_mae[0]:.4f}, ValP@5: {val_mae[1]:.4f}")
            print(time.time() - t0)
            test_idx_list, _ = self.test(self.train_idx)
            # the reference pickles the whole module (:147; un-loadable without its class files): the state_dict here
            torch.save(self.model.state_dict(), os.path.join(out_dir, f"model-{epoch}.pkl"))
            with open(os.path.join(out_dir, f"testServices-epoch{epoch}.txt"), "w") as f:
                json.dump(test_idx_list + val_idx_list, f)                          # :148-149
        return self.model


class _Data:
    pass


def _as_data(b):
    """The PyG-style object Net.forward takes, over an MLBatch's tensors (evaluation path of TrainML.test)."""
    d = _Data()
    d.x = b.x
    d._gnnpn_csr = {"wf": b.wf, "seg": b.seg, "svc": b.svc, "xs": b.xs, "copies": b.B, "max_nodes": 0,
                    "mean_rowptr": b.mean_rowptr, "mean_col": b.mean_col}
    return d
