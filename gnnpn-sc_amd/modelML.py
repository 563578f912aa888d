"""GNN candidate-ranking model of the ML+2PN path behind the reference's entry point
(/root/reference/src/models/modelML.py): ``Net(hiddenChannels, outChannels, embeddingChannels,
numLayersGIN, numLayersGCN, isServices=True, dropout=0.0)`` with ``forward(data) -> [B,S]`` sigmoid
scores, same ``state_dict`` keys/shapes (``nodeConvs.i.{eps,nn.0,nn.1,nn.3}``, ``serviceConvs.i.
{weight [in,out], bias}``, ...).  torch modules are parameter containers only; the arithmetic runs in
libgnnpn_hip.so:

  GIN  (workflow graph) : csr_aggregate(sum_j x_j + (1+eps) x_i) -> linear+BN+ReLU -> linear+BN+ReLU
  GCN  (service graph)  : gcn_norm -> linear (X.W, transform first) -> csr_aggregate(+bias,BN,ReLU)
  head                  : linear, segment_mean, linear(score) + sigmoid

Service-branch semantics: ``Net.forward(data)`` does what the reference's forward does on
whatever the batched ``data`` holds (GCN over all copies, mean over copies: modelML.py:145-156,167-172); the device
pipeline (``pipeline.ML2PNPipeline``) evaluates the branch on ONE copy of the table — the problem-independent
embedding — once per (weights, table).
"""
import os

import torch
from torch import nn

from . import custom_ops, graph, ops   # noqa: F401  (custom_ops registers torch.ops.gnnpn.*)
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID

BN_EPS = 1e-5


class NodeEncoder(nn.Module):
    """modelML.py:9-29 — nine Embedding(vocab, c) tables, of which callers only ever use table 0
    (they pass one column, :134,146).  ``vocab`` defaults to the reference's 100 (:16)."""

    def __init__(self, hiddenChannels, vocab=100):
        super().__init__()
        self.embeddings = nn.ModuleList(nn.Embedding(vocab, hiddenChannels) for _ in range(9))

    def reset_parameters(self):
        for e in self.embeddings:
            e.reset_parameters()


class GINConv(nn.Module):
    """Parameter container with torch_geometric 1.7.0's GINConv(nn, train_eps=True) layout."""

    def __init__(self, mlp):
        super().__init__()
        self.nn = mlp
        self.eps = nn.Parameter(torch.zeros(1))

    def reset_parameters(self):
        for m in self.nn:
            if hasattr(m, "reset_parameters"):
                m.reset_parameters()
        self.eps.data.zero_()


class GCNConv(nn.Module):
    """Parameter container with torch_geometric 1.7.0's GCNConv layout (weight stored in x out)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight)
        nn.init.zeros_(self.bias)


def _bn_affine(bn):
    """Eval-mode BatchNorm1d as y = x*alpha + beta, computed the way ATen's CPU kernel does
    (alpha = weight * 1/sqrt(var+eps); beta = bias - mean*alpha), in fp32 on the host, once."""
    var, mean = bn.running_var.detach().float().cpu(), bn.running_mean.detach().float().cpu()
    w, b = bn.weight.detach().float().cpu(), bn.bias.detach().float().cpu()
    alpha = w * (1.0 / torch.sqrt(var + bn.eps))
    beta = b - mean * alpha
    return alpha, beta


class Net(nn.Module):
    def __init__(self, hiddenChannels, outChannels, embeddingChannels, numLayersGIN, numLayersGCN,
                 isServices=True, dropout=0.0, vocab=100):
        super().__init__()
        self.numLayersGIN, self.numLayersGCN = numLayersGIN, numLayersGCN
        self.dropout, self.outChannels = dropout, outChannels
        self.reqAndServiceChannels = embeddingChannels
        self.qosNumber, self.constraintNumber, self.isService = 4, 2, isServices
        h, c = hiddenChannels, embeddingChannels

        self.nodeEncoder = NodeEncoder(c, vocab)
        self.serviceEncoder = NodeEncoder(c, vocab)
        self.nodeConvs, self.nodeBatchNorms = nn.ModuleList(), nn.ModuleList()
        for i in range(numLayersGIN):                                              # :75-92
            in_f = c + self.constraintNumber * 3 if i == 0 else h
            self.nodeConvs.append(GINConv(nn.Sequential(nn.Linear(in_f, 2 * h), nn.BatchNorm1d(2 * h), nn.ReLU(),
                                                        nn.Linear(2 * h, h))))
            self.nodeBatchNorms.append(nn.BatchNorm1d(h))
        self.nodeLin = nn.Linear(h, h)                                             # :93
        self.serviceConvs, self.serviceBatchNorms = nn.ModuleList(), nn.ModuleList()
        for i in range(numLayersGCN):                                              # :98-104
            self.serviceConvs.append(GCNConv(c + self.qosNumber if i == 0 else 2 * h, 2 * h))
            self.serviceBatchNorms.append(nn.BatchNorm1d(2 * h))
        self.serviceLin = nn.Linear(2 * h, h)                                      # :106
        self.noServicesLins = nn.ModuleList(                                       # :108-115 (isServices=False branch)
            nn.Linear(c + self.qosNumber if i == 0 else 2 * h, 2 * h) for i in range(numLayersGCN))
        self._prep = None
        self.fuse_request_branch = os.environ.get("GNNPN_LAYERED_GIN") != "1"   # one-launch GIN branch for small workflow graphs
        self.fuse_gin_layers = os.environ.get("GNNPN_LAYERED_GIN") != "1"       # one launch per GIN layer for large ones (same bits)
        # arithmetic of the dense products of those large layers: "f32" (the matrix core's fp32 chain, bit-identical to the layered
        # kernels) or "split" (fp16 matrix cores through the exact 3-piece split, gin_layer_split.hip).  ML2PNPipeline passes its
        # own precision per call; this is the default of direct calls (Net.forward, TrainML.test).
        self.dense_precision = os.environ.get("GNNPN_DENSE_PRECISION", "f32")
        self.parallel_branches = os.environ.get("GNNPN_SERIAL_BRANCHES") != "1"   # scores(): GCN branch on a side stream
        self._side_streams = {}

    def reset_parameters(self):                                                    # :117-129
        self.nodeEncoder.reset_parameters()
        self.serviceEncoder.reset_parameters()
        for conv, bn in zip(self.nodeConvs, self.nodeBatchNorms):
            conv.reset_parameters()
            bn.reset_parameters()
        self.nodeLin.reset_parameters()
        for conv, bn in zip(self.serviceConvs, self.serviceBatchNorms):
            conv.reset_parameters()
            bn.reset_parameters()
        self.serviceLin.reset_parameters()
        self._prep = None

    def _load_from_state_dict(self, *a, **k):
        self._prep = None
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._prep = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._prep = None
        return super().load_state_dict(*a, **k)

    def prepared(self, device):
        """Kernel-ready constants (BN folded to alpha/beta, GCN weights as [out,in]) on ``device``."""
        if self._prep is not None and self._prep["device"] == device:
            return self._prep
        f = lambda t: t.detach().float().contiguous().to(device)   # noqa: E731
        p = {"device": device, "gin": [], "gcn": [],
             "node_table": f(self.nodeEncoder.embeddings[0].weight),
             "service_table": f(self.serviceEncoder.embeddings[0].weight)}
        for conv, bn in zip(self.nodeConvs, self.nodeBatchNorms):
            a1, b1 = _bn_affine(conv.nn[1])
            a2, b2 = _bn_affine(bn)
            p["gin"].append({"eps": f(conv.eps), "w0": f(conv.nn[0].weight), "b0": f(conv.nn[0].bias),
                             "a1": a1.to(device), "s1": b1.to(device), "w3": f(conv.nn[3].weight),
                             "b3": f(conv.nn[3].bias), "a2": a2.to(device), "s2": b2.to(device)})
            if conv.nn[3].weight.shape[0] % 16 == 0:      # MFMA B-fragment layout for the one-launch GIN branch
                p["gin"][-1].update(w0p=ops.pack_mfma_b(p["gin"][-1]["w0"]), w3p=ops.pack_mfma_b(p["gin"][-1]["w3"]))
            if conv.nn[0].weight.shape[0] % 32 == 0 and conv.nn[3].weight.shape[0] % 32 == 0:   # ... and for the one-launch layer
                p["gin"][-1].update(w0q=ops.pack_mfma_b32(p["gin"][-1]["w0"]), w3q=ops.pack_mfma_b32(p["gin"][-1]["w3"]))
                if device.type == "cuda":    # ... and split exactly into fp16 pieces for its fp16-matrix-core form
                    p["gin"][-1].update(w0s=ops.pack_split_weights(p["gin"][-1]["w0"]), w3s=ops.pack_split_weights(p["gin"][-1]["w3"]))
        for conv, bn in zip(self.serviceConvs, self.serviceBatchNorms):
            a, b = _bn_affine(bn)
            p["gcn"].append({"wt": f(conv.weight.detach().t()), "bias": f(conv.bias), "a": a.to(device),
                             "s": b.to(device)})
        p["nolin"] = [{"w": f(lin.weight), "bias": f(lin.bias), "a": g["a"], "s": g["s"]}
                      for lin, g in zip(self.noServicesLins, p["gcn"])]
        p["nodeLin"] = (f(self.nodeLin.weight), f(self.nodeLin.bias))
        if self.nodeLin.weight.shape[0] % 16 == 0:
            p["nodeLin_p"] = ops.pack_mfma_b(p["nodeLin"][0])
        if self.nodeLin.weight.shape[0] % 32 == 0:
            p["nodeLin_q"] = ops.pack_mfma_b32(p["nodeLin"][0])
            if device.type == "cuda":
                p["nodeLin_s"] = ops.pack_split_weights(p["nodeLin"][0])
        p["serviceLin"] = (f(self.serviceLin.weight), f(self.serviceLin.bias))
        self._prep = p
        return p

    # ---- the two branches + head, on kernel-ready inputs ---------------------------------------
    @torch.no_grad()
    def fused_request_branch_ok(self, x, max_nodes):
        """Shapes gnnpn_request_branch_f32 is built for (else the separate kernels; same bits either way)."""
        h = self.nodeLin.weight.shape[0]
        return (self.fuse_request_branch and 0 < max_nodes <= ops.REQUEST_BRANCH_MAX_NODES and h == 128 and
                self.numLayersGIN <= 4 and self.reqAndServiceChannels + x.shape[1] - 1 <= 32)

    @torch.no_grad()
    def request_embedding(self, x, wf_csr, seg_ptr, max_nodes=0, dense_precision=None):
        """Workflow branch (modelML.py:133-143,165-166): x [N,7], CSR of the batched workflow graphs,
        graph segment pointer -> [B, hidden].  ``max_nodes`` > 0 promises that every graph has at most that many nodes
        and that every edge stays inside its graph: up to 16 the whole branch then runs as ONE launch."""
        p = self.prepared(x.device)
        if self.fused_request_branch_ok(x, max_nodes):
            flat = [lp[k] for lp in p["gin"] for k in ("w0p", "b0", "a1", "s1", "w3p", "b3", "a2", "s2", "eps")]
            return torch.ops.gnnpn.request_branch(x, p["node_table"], wf_csr.rowptr, wf_csr.col, seg_ptr, int(max_nodes),
                                                  flat, p["nodeLin_p"], p["nodeLin"][1], 128)
        h = torch.ops.gnnpn.embed_concat(x, p["node_table"])                                               # :134-137
        n_lin = self.nodeLin.weight.shape[0]
        fused = self.fuse_gin_layers and h.shape[0] >= 4096 and all(
            "w0q" in lp and ops.gin_layer_supported(lp["w0"].shape[1], lp["w0"].shape[0], lp["w3"].shape[0]) for lp in p["gin"])
        last = len(p["gin"]) - 1
        lin_done = False
        prec = self.dense_precision if dense_precision is None else dense_precision
        if prec not in ("f32", "split"):
            raise ops.GnnpnError(f"dense_precision must be 'f32' or 'split', got {prec!r}")
        split = fused and prec == "split" and all("w0s" in lp for lp in p["gin"])
        for i, lp in enumerate(p["gin"]):                                                       # :139-142
            if fused:    # one launch per layer (the [rows x 256] intermediate stays on the CU), nodeLin behind the last: same bits
                with_lin = i == last and "nodeLin_q" in p and ops.gin_layer_supported(lp["w0"].shape[1], 256, 128, n_lin)
                if split and ops.gin_layer_split_supported(lp["w0"].shape[1], 256, 128):
                    with_lin = with_lin and "nodeLin_s" in p
                    h = torch.ops.gnnpn.gin_layer_split(wf_csr.rowptr, wf_csr.col, h, lp["eps"], *lp["w0s"], lp["b0"], lp["a1"], lp["s1"],
                                                        *lp["w3s"], lp["b3"], lp["a2"], lp["s2"],
                                                        *(p["nodeLin_s"] if with_lin else (None, None)),
                                                        p["nodeLin"][1] if with_lin else None)
                    lin_done = with_lin
                    continue
                h = torch.ops.gnnpn.gin_layer(wf_csr.rowptr, wf_csr.col, h, lp["eps"], lp["w0q"], lp["b0"], lp["a1"], lp["s1"],
                                              lp["w3q"], lp["b3"], lp["a2"], lp["s2"],
                                              p["nodeLin_q"] if with_lin else None, p["nodeLin"][1] if with_lin else None)
                lin_done = with_lin
                continue
            agg = torch.ops.gnnpn.csr_aggregate(wf_csr.rowptr, wf_csr.col, None, h, self_coef=lp["eps"])
            t = torch.ops.gnnpn.linear(agg, lp["w0"], lp["b0"], lp["a1"], lp["s1"], ACT_RELU)
            h = torch.ops.gnnpn.linear(t, lp["w3"], lp["b3"], lp["a2"], lp["s2"], ACT_RELU)
        if not lin_done:
            h = torch.ops.gnnpn.linear(h, *p["nodeLin"])                                                    # :165
        return torch.ops.gnnpn.segment_mean(seg_ptr, h)                                                     # :166

    @torch.no_grad()
    def service_embedding(self, x_service, svc_csr):
        """Service branch (modelML.py:145-156,164): x_service [S,5], self-loop-complete CSR with RAW
        edge weights -> [S, hidden]."""
        p = self.prepared(x_service.device)
        xs = torch.ops.gnnpn.embed_concat(x_service, p["service_table"])                                    # :146-149
        if not self.isService:                                                                  # :157-162 graph-free ablation
            for lp in p["nolin"]:
                xs = torch.ops.gnnpn.linear(xs, lp["w"], lp["bias"], lp["a"], lp["s"], ACT_RELU)
            return torch.ops.gnnpn.linear(xs, *p["serviceLin"])                                             # :164
        # the normalised edge weights are a function of the graph alone: computed once per CSR object (GCNConv recomputes the
        # same values in every forward) — which also keeps the tensor, and with it the aggregate's per-(graph, weights) plan
        # (ops.csr_tile_plan), the same from call to call
        norm = getattr(svc_csr, "_gcn_norm", None)
        if norm is None:
            norm = torch.ops.gnnpn.gcn_norm(svc_csr.rowptr, svc_csr.col, svc_csr.w)
            if not torch.cuda.is_current_stream_capturing():      # (memory of a capture's private pool must not outlive the graph)
                svc_csr._gcn_norm = norm
        for lp in p["gcn"]:                                                                     # :152-155
            xw = torch.ops.gnnpn.linear(xs, lp["wt"])                                                       # transform first
            xs = custom_ops.csr_aggregate(svc_csr.rowptr, svc_csr.col, norm, xw, bias=lp["bias"], scale=lp["a"],
                                          shift=lp["s"], act=ACT_RELU, block_rows=svc_csr.block_rows)   # form chosen per graph
        return torch.ops.gnnpn.linear(xs, *p["serviceLin"])                                                 # :164

    @torch.no_grad()
    def scores(self, x, wf_csr, seg_ptr, x_service, svc_csr, service_emb=None, max_nodes=0, dense_precision=None):
        """The two branches are independent until the score product: the service branch (GCN) is forked onto
        a side stream and joined before the GEMM, so ~20 small launch-latency-bound kernels run two abreast
        (fork/join is stream-capturable: inside a HIP graph it becomes two parallel branches).
        ``service_emb`` [S, hidden]: the service branch's result computed earlier (it depends on the weights and the
        service table only — pipeline.ML2PNPipeline caches it per (weights, table)); the branch is then skipped."""
        if service_emb is not None:
            xr = self.request_embedding(x, wf_csr, seg_ptr, max_nodes, dense_precision)
            return torch.ops.gnnpn.linear(xr, service_emb, act=ACT_SIGMOID)                                 # :173-176
        if not self.parallel_branches:
            xr = self.request_embedding(x, wf_csr, seg_ptr, max_nodes, dense_precision)
            xs = self.service_embedding(x_service, svc_csr)
            return torch.ops.gnnpn.linear(xr, xs, act=ACT_SIGMOID)                                          # :173-176
        cur = torch.cuda.current_stream(x.device)
        side = self._side_streams.get(x.device)
        if side is None:
            side = self._side_streams[x.device] = torch.cuda.Stream(x.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            xs = self.service_embedding(x_service, svc_csr)
        xr = self.request_embedding(x, wf_csr, seg_ptr, max_nodes, dense_precision)
        cur.wait_stream(side)
        xs.record_stream(cur)
        return torch.ops.gnnpn.linear(xr, xs, act=ACT_SIGMOID)                                              # :173-176

    def forward(self, data):
        """Net.forward (modelML.py:131-176) on a PyG-style ``data`` object with attributes x,
        edge_index, batch, x_service, edge_index_service, edge_attr_service (device tensors).
        The CSR layouts are cached on ``data`` (``_gnnpn_csr``) so repeated forwards skip the sort.

        Service branch, exactly as the reference evaluates it: the GCN layers run over ALL rows of ``x_service`` with
        ``edge_index_service`` AS GIVEN — a batch of B graphs carries B concatenated copies of the service table
        (trainML.py:109-114) and whatever index offsets the batching gave copy b's edges (torch_geometric 1.7.0 shifts
        them by the workflow node counts, see oracle/ml.py) — and ``scatter(.., serviceBatch, reduce='mean')``
        (:167-172) then averages row s of every copy.  No layout is assumed and nothing is masked away.  With ONE copy
        (``x_service`` has outChannels rows) this is the problem-independent embedding the device pipeline caches."""
        if self.training:
            raise NotImplementedError("Net.forward is the inference forward; the training step (BatchNorm on batch statistics, "
                                      "BCE, backward, Adam) is gnnpn_sc_amd.trainML.ml_train_step / TrainML.train")
        x = data.x.squeeze().float().contiguous()
        cache = getattr(data, "_gnnpn_csr", None)
        if cache is None:
            S = self.outChannels
            n_graphs = int(data.batch.max().item()) + 1
            xs = data.x_service.squeeze().float().contiguous()
            n_svc = xs.shape[0]
            if n_svc != S and n_svc != n_graphs * S:      # the reference's scatter would fail on the size mismatch (:167-172)
                raise ValueError(f"x_service has {n_svc} rows: expected outChannels = {S} (one copy) or one copy per "
                                 f"graph = {n_graphs * S}")
            copies = n_svc // S
            b = data.batch.long()
            inside = bool((b[data.edge_index[0].long()] == b[data.edge_index[1].long()]).all().item()) if data.edge_index.numel() else True
            cache = {"wf": graph.csr_by_destination(data.edge_index, x.shape[0]),
                     "seg": graph.segment_ptr(data.batch, n_graphs),
                     "svc": graph.gcn_csr(data.edge_index_service, data.edge_attr_service, n_svc),
                     "xs": xs, "copies": copies,
                     "max_nodes": int(torch.bincount(b).max().item()) if inside else 0}
            if graph.block_local(data.edge_index_service, S):     # clean block-diagonal copies: features staged in LDS
                cache["svc"].block_rows = S
            if copies > 1:    # rows (s, S+s, 2S+s, ...) as one CSR row: csr_aggregate sums them in copy order = scatter order
                dev = xs.device
                cache["mean_rowptr"] = (torch.arange(S + 1, device=dev, dtype=torch.int32) * copies).contiguous()
                cache["mean_col"] = (torch.arange(S, device=dev, dtype=torch.int32).view(S, 1) +
                                     S * torch.arange(copies, device=dev, dtype=torch.int32).view(1, copies)).reshape(-1).contiguous()
            try:
                data._gnnpn_csr = cache
            except AttributeError:
                pass
        if cache["copies"] == 1:
            return self.scores(x, cache["wf"], cache["seg"], cache["xs"], cache["svc"], max_nodes=cache["max_nodes"])
        emb_all = self.service_embedding(cache["xs"], cache["svc"])                            # :145-156,164
        total = torch.ops.gnnpn.csr_aggregate(cache["mean_rowptr"], cache["mean_col"], None, emb_all)
        emb = total / float(cache["copies"])                                                    # :172 (sum / count)
        return self.scores(x, cache["wf"], cache["seg"], None, None, service_emb=emb, max_nodes=cache["max_nodes"])
