"""Stand-ins for the two third-party packages /root/reference/src/models/modelML.py imports
(``torch_geometric==1.7.0``: GINConv, GCNConv; ``torch_scatter==2.0.6``: scatter —
requirements.txt:6-7) which are NOT in /root/reference and NOT installed in this image.

Used ONLY by tests/golden/make_golden.py (build container) so that the reference's own
``Net.__init__`` / ``Net.forward`` glue can run unmodified.  They restate the packages' published
message-passing algorithm (gather x_j by edge_index[0], message, scatter-add onto edge_index[1]);
they are NOT the packages, so the conv arithmetic stays "parity unpinned" (oracle/__init__.py).
Parameter names match PyG 1.7.0 so that ``Net.state_dict()`` has the reference's keys.
"""
import sys
import types

import torch
from torch import nn


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    assert dim == 0
    n = int(index.max()) + 1 if dim_size is None else dim_size
    shape = (n,) + tuple(src.shape[1:])
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    total = torch.zeros(shape, dtype=src.dtype).scatter_add_(0, idx, src)
    if reduce in ("sum", "add"):
        return total
    if reduce == "mean":
        ones = torch.ones(index.numel(), dtype=src.dtype)
        count = torch.zeros(n, dtype=src.dtype).scatter_add_(0, index, ones).clamp_(min=1)
        return total.true_divide(count.view(-1, *([1] * (src.dim() - 1))))
    raise NotImplementedError(reduce)


class _MessagePassing(nn.Module):
    """aggr='add', flow source_to_target: x_j = x[edge_index[0]], aggregated at edge_index[1]."""

    def propagate(self, edge_index, x, **kw):
        msg = self.message(x[edge_index[0]], **kw)
        return scatter(msg, edge_index[1], dim=0, dim_size=x.size(0), reduce="sum")


class GINConv(_MessagePassing):
    def __init__(self, nn_module, eps=0.0, train_eps=False):
        super().__init__()
        self.nn = nn_module
        self.initial_eps = eps
        if train_eps:
            self.eps = nn.Parameter(torch.Tensor([eps]))
        else:
            self.register_buffer("eps", torch.Tensor([eps]))

    def reset_parameters(self):
        for m in self.nn:
            if hasattr(m, "reset_parameters"):
                m.reset_parameters()
        self.eps.data.fill_(self.initial_eps)

    def message(self, x_j):
        return x_j

    def forward(self, x, edge_index):
        out = self.propagate(edge_index, x)
        out += (1 + self.eps) * x
        return self.nn(out)


class GCNConv(_MessagePassing):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.weight = nn.Parameter(torch.Tensor(in_channels, out_channels))
        self.bias = nn.Parameter(torch.Tensor(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        bound = (6.0 / (self.weight.size(0) + self.weight.size(1))) ** 0.5
        self.weight.data.uniform_(-bound, bound)
        self.bias.data.zero_()

    @staticmethod
    def _norm(edge_index, edge_weight, n):
        row, col = edge_index
        mask = row != col
        loop_index = torch.arange(n, dtype=row.dtype).unsqueeze(0).repeat(2, 1)
        loop_weight = torch.full((n,), 1.0, dtype=edge_weight.dtype)
        rest = edge_weight[~mask]
        if rest.numel() > 0:
            loop_weight[row[~mask]] = rest
        edge_index = torch.cat([edge_index[:, mask], loop_index], dim=1)
        edge_weight = torch.cat([edge_weight[mask], loop_weight], dim=0)
        row, col = edge_index
        deg = scatter(edge_weight, col, dim=0, dim_size=n, reduce="sum")
        deg_inv_sqrt = deg.pow_(-0.5)
        deg_inv_sqrt.masked_fill_(deg_inv_sqrt == float("inf"), 0)
        return edge_index, deg_inv_sqrt[row] * edge_weight * deg_inv_sqrt[col]

    def message(self, x_j, edge_weight):
        return edge_weight.view(-1, 1) * x_j

    def forward(self, x, edge_index, edge_weight):
        edge_index, edge_weight = self._norm(edge_index, edge_weight, x.size(0))
        x = torch.matmul(x, self.weight)
        out = self.propagate(edge_index, x, edge_weight=edge_weight)
        out += self.bias
        return out


def install():
    """Register the stand-ins under the names modelML.py:5-6 imports."""
    tg = types.ModuleType("torch_geometric")
    tg_nn = types.ModuleType("torch_geometric.nn")
    tg_nn.GINConv, tg_nn.GCNConv = GINConv, GCNConv
    tg.nn = tg_nn
    ts = types.ModuleType("torch_scatter")
    ts.scatter = scatter
    sys.modules.update({"torch_geometric": tg, "torch_geometric.nn": tg_nn, "torch_scatter": ts})
