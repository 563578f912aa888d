"""Graph layout helpers: edge lists (the reference's ``edge_index`` [2,E]) -> destination-major
CSR with the edge order preserved inside every row, which is what makes the gather-aggregate
kernel reproduce scatter_add's summation order.  Pure data movement (sort / cumsum) — done once
per graph with torch ops on whatever device the edge list lives on, no feature arithmetic.
"""
from dataclasses import dataclass

import torch


@dataclass
class CSR:
    rowptr: torch.Tensor        # int32 [n+1]
    col: torch.Tensor           # int32 [nnz]   source node of every in-edge
    w: torch.Tensor             # float32 [nnz] or None
    n: int
    block_rows: int = 0         # > 0: every edge stays inside its block of that many consecutive rows (block_local())

    def to(self, device):
        return CSR(self.rowptr.to(device), self.col.to(device), None if self.w is None else self.w.to(device),
                   self.n, self.block_rows)


def csr_by_destination(edge_index, n, weight=None):
    """Messages flow edge_index[0] -> edge_index[1] (PyG source_to_target).  Stable sort by
    destination keeps each row's in-edges in edge-list order."""
    src, dst = edge_index[0].long(), edge_index[1].long()
    order = torch.sort(dst, stable=True).indices
    counts = torch.bincount(dst, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int32, device=dst.device)
    rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    w = None if weight is None else weight.float()[order].contiguous()
    return CSR(rowptr.contiguous(), src[order].to(torch.int32).contiguous(), w, n)


def gcn_csr(edge_index, edge_weight, n):
    """add_remaining_self_loops(fill_value=1) of torch_geometric 1.7.0 (invoked by GCNConv.forward,
    call site /root/reference/src/models/modelML.py:153) as an edge-list edit — existing self loops
    keep their weight, every other node gets a weight-1 loop, ALL loops go after the non-loop
    edges — followed by the CSR layout.  The symmetric normalisation itself is arithmetic and runs
    in the kernel gnnpn_gcn_norm_f32."""
    row, col = edge_index[0].long(), edge_index[1].long()
    keep = row != col
    loop_w = torch.ones(n, dtype=torch.float32, device=row.device)
    if bool((~keep).any()):
        loop_w[row[~keep]] = edge_weight.float()[~keep]
    loops = torch.arange(n, dtype=torch.long, device=row.device)
    ei = torch.stack([torch.cat([row[keep], loops]), torch.cat([col[keep], loops])])
    return csr_by_destination(ei, n, torch.cat([edge_weight.float()[keep], loop_w]))


def block_local(edge_index, block_rows):
    """True when every edge's two endpoints lie in the same block of ``block_rows`` consecutive node ids — a batch of
    block-diagonal graph copies.  The LDS-staged aggregate (gnnpn_csr_aggregate_blocks_f32) relies on it."""
    if edge_index.numel() == 0:
        return True
    return bool((torch.div(edge_index[0], block_rows, rounding_mode="floor") ==
                 torch.div(edge_index[1], block_rows, rounding_mode="floor")).all().item())


def segment_ptr(batch, n_graphs):
    """PyG ``batch`` vector (graph id per node, nodes of a graph contiguous) -> int32 [B+1]."""
    counts = torch.bincount(batch.long(), minlength=n_graphs)
    ptr = torch.zeros(n_graphs + 1, dtype=torch.int32, device=batch.device)
    ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    if not bool((batch[1:] >= batch[:-1]).all()):
        raise ValueError("batch vector must be sorted (nodes of a graph contiguous)")
    return ptr
