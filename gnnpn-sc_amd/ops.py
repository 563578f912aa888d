"""Thin tensor-level wrappers over the C ABI (``include/gnnpn_hip.h``): validate, allocate the
output with torch (device memory only), pass raw pointers + the current HIP stream.  No compute
happens in Python; there is no CPU path — non-CUDA tensors raise ``GnnpnError``.
"""
import itertools
import weakref

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, GnnpnError, check, dev_ptr, ptr_array, stream_ptr  # noqa: F401

F32, I32, F64, U8 = torch.float32, torch.int32, torch.float64, torch.uint8


def _rows2d(t, name):
    if t.dim() != 2:
        raise GnnpnError(f"{name}: expected a 2-D tensor, got shape {tuple(t.shape)}")
    return t


def linear(a, weight, bias=None, scale=None, shift=None, act=ACT_NONE, out=None):
    """out[M,N] = act((a[M,K] @ weight[N,K]^T + bias) * scale + shift)   (gnnpn_linear_f32)."""
    a, weight = _rows2d(a, "linear.a"), _rows2d(weight, "linear.weight")
    M, K = a.shape
    N = weight.shape[0]
    if weight.shape[1] != K:
        raise GnnpnError(f"linear: K mismatch {a.shape} x {weight.shape}")
    if out is None:
        out = torch.empty((M, N), dtype=F32, device=a.device)
    lib = _lib.load()
    check(lib.gnnpn_linear_f32(dev_ptr(a, F32, "a"), K, dev_ptr(weight, F32, "weight"), K,
                               dev_ptr(bias, F32, "bias", True), dev_ptr(scale, F32, "scale", True),
                               dev_ptr(shift, F32, "shift", True), act, dev_ptr(out, F32, "out"), N, M, N, K,
                               stream_ptr()), "gnnpn_linear_f32")
    return out


def embed_concat(x, table):
    """[n, 1+f] rows ``[id, f floats]`` -> [n, emb+f] = [table[id] | floats]   (gnnpn_embed_concat_f32)."""
    x, table = _rows2d(x, "embed_concat.x"), _rows2d(table, "embed_concat.table")
    n, nfeat = x.shape[0], x.shape[1] - 1
    vocab, emb = table.shape
    out = torch.empty((n, emb + nfeat), dtype=F32, device=x.device)
    check(_lib.load().gnnpn_embed_concat_f32(dev_ptr(x, F32, "x"), dev_ptr(table, F32, "table"), vocab, emb, nfeat,
                                             dev_ptr(out, F32, "out"), n, stream_ptr()), "gnnpn_embed_concat_f32")
    return out


LDS_BLOCK_ROWS_MAX = 10239      # (block_rows + 1) * 16 B <= 160 KB: the block and one all-zero row
LDS_MIN_WORKGROUPS = 128        # below this many (block, slice) workgroups the one-wave-per-row gather fills the chip better
LDS_SLICE16_ROWS_MAX = 2559     # blocks up to here stage 16-channel slices: the LDS-staged form is the faster one there
PREFER_LDS_AGGREGATE = None     # None: the LDS-staged form where it is measured faster (16-channel slices, enough workgroups:
#                                 profiles/r02_csr_aggregate_roofline.json); True / False force the choice
_row_orders = {}                # id(rowptr tensor) -> (weak reference to it, block_rows, order)


def csr_block_row_order(rowptr, block_rows):
    """Per block of ``block_rows`` rows: the rows by descending edge count (gnnpn_csr_block_row_order) — the order in which
    the LDS-staged aggregate deals a block's rows to its wavefronts.  Cached per rowptr tensor (a property of the graph)."""
    key = id(rowptr)
    hit = _row_orders.get(key)
    if hit is not None and hit[0]() is rowptr and hit[1] == block_rows:
        return hit[2]
    n = rowptr.numel() - 1
    order = torch.empty(n, dtype=I32, device=rowptr.device)
    check(_lib.load().gnnpn_csr_block_row_order(dev_ptr(rowptr, I32, "rowptr"), n, int(block_rows), dev_ptr(order, I32, "order"),
                                                stream_ptr()), "gnnpn_csr_block_row_order")
    if not torch.cuda.is_current_stream_capturing():      # memory of a capture's private pool must not outlive the graph
        _row_orders[key] = (weakref.ref(rowptr, lambda _, k=key: _row_orders.pop(k, None)), block_rows, order)
    return order


PREFER_TILED_AGGREGATE = None   # None: the tiled form (destination tile x source tile, one source tile's slice staged in LDS at a
#                                 time, edge lists from the plan's stream) for block-local graphs whose plan is valid — rows in
#                                 source-tile order, the reference's own edge order — and that have enough (block, destination
#                                 tile, slice) workgroups to fill the chip: measured faster than both other forms from 2507 to
#                                 20000 rows per block (profiles/r04_csr_aggregate_roofline.json); True: wherever the plan is
#                                 valid; False: never
TILED_MIN_WORKGROUPS = 128
_tile_plans = {}                # (id(rowptr), id(col), id(w), block_rows) -> (weak references, TilePlan)


class TilePlan:
    """The plan of gnnpn_csr_aggregate_tiled_f32 for one (graph, weights): tile geometry, the per-unit headers, the row
    order, the trailing self loops and the sliced-ELL stream of (LDS offset, weight) batches — built once (three launches
    and one host read of the batch count) and reused by every layer and call.  ``valid`` is False when some row's
    neighbour list does not visit the source tiles in order (the sums would be taken in another order): callers then
    keep the gather form.  ``stats``: edges, (row, edge) slots in the stream, its efficiency and the histogram of the
    per-(row, source tile) run lengths."""

    def __init__(self, rowptr, col, w, block_rows):
        import ctypes
        lib = _lib.load()
        n = rowptr.numel() - 1
        g = _lib.TilePlanGeom()
        self.valid, self.block_rows, self.n_rows = False, int(block_rows), n
        self.stats = {}
        if n <= 0 or lib.gnnpn_csr_tile_plan_geometry(n, int(block_rows), ctypes.byref(g)) != 0:
            return
        dev = rowptr.device
        self.geom = {k: int(getattr(g, k)) for k in ("n_blocks", "src_tiles", "src_tile_rows", "dst_tiles", "dst_tile_rows", "units",
                                                     "wavefronts", "passes")}
        self.header = torch.empty(g.header_bytes // 4, dtype=I32, device=dev)
        self.order = torch.empty(g.order_bytes // 4, dtype=I32, device=dev)
        tstart = torch.empty(g.tstart_bytes // 4, dtype=I32, device=dev)
        self.selfw = torch.empty(g.selfw_bytes // 4, dtype=F32, device=dev)
        meta = torch.empty(g.meta_bytes // 4, dtype=I32, device=dev)
        check(lib.gnnpn_csr_tile_plan_rows(dev_ptr(rowptr, I32, "rowptr"), dev_ptr(col, I32, "col"), dev_ptr(w, F32, "w", True), n,
                                           int(block_rows), dev_ptr(self.header, I32, "header"), dev_ptr(self.order, I32, "order"),
                                           dev_ptr(tstart, I32, "tstart"), dev_ptr(self.selfw, F32, "selfw"),
                                           dev_ptr(meta, I32, "meta"), stream_ptr()), "gnnpn_csr_tile_plan_rows")
        m = [int(v) & 0xFFFFFFFF for v in meta.cpu().tolist()]          # the one host read: validity and the stream's size
        hist = m[8:72]
        self.stats = {"invalid_rows": m[0], "quads": m[1], "edges": m[2], "slots": m[3], "rows": m[4],
                      "efficiency": (m[2] / m[3]) if m[3] else 1.0, "stream_bytes": m[1] * 512, "run_histogram": hist,
                      "mean_run": (sum(i * h for i, h in enumerate(hist)) / max(1, sum(hist)))}
        if m[0] != 0:
            return
        self.batches = torch.zeros((m[1] + 4) * 512, dtype=U8, device=dev)      # + FOUR quads of slack: every unit's first four quads are requested, also an empty last unit's (first == quads)
        check(lib.gnnpn_csr_tile_plan_fill(dev_ptr(col, I32, "col"), dev_ptr(w, F32, "w", True), n, int(block_rows),
                                           dev_ptr(self.header, I32, "header"), dev_ptr(self.order, I32, "order"),
                                           dev_ptr(tstart, I32, "tstart"), dev_ptr(self.batches, U8, "batches"), m[1], stream_ptr()),
              "gnnpn_csr_tile_plan_fill")
        self.valid = True

    def aggregate(self, x, self_coef=None, bias=None, scale=None, shift=None, act=ACT_NONE):
        x = _rows2d(x, "csr_aggregate_tiled.x")
        C = x.shape[1]
        y = torch.empty((self.n_rows, C), dtype=F32, device=x.device)
        check(_lib.load().gnnpn_csr_aggregate_tiled_f32(
            dev_ptr(self.header, I32, "header"), dev_ptr(self.order, I32, "order"), dev_ptr(self.selfw, F32, "selfw"),
            dev_ptr(self.batches, U8, "batches"), dev_ptr(x, F32, "x"), C, dev_ptr(self_coef, F32, "self_coef", True),
            dev_ptr(bias, F32, "bias", True), dev_ptr(scale, F32, "scale", True), dev_ptr(shift, F32, "shift", True), act,
            dev_ptr(y, F32, "y"), C, self.n_rows, C, self.block_rows, stream_ptr()), "gnnpn_csr_aggregate_tiled_f32")
        return y


def csr_tile_plan(rowptr, col, w, block_rows):
    """The cached TilePlan of (rowptr, w) — a property of the graph and its weights, like csr_block_row_order — or None
    while a stream capture is in progress and none has been built yet (building reads the batch count back)."""
    key = (id(rowptr), id(col), id(w) if w is not None else 0, int(block_rows))
    hit = _tile_plans.get(key)
    if hit is not None and hit[0]() is rowptr and hit[3]() is col and (w is None or hit[1]() is w):
        return hit[2]
    if torch.cuda.is_current_stream_capturing():
        return None
    plan = TilePlan(rowptr, col, w, block_rows)
    drop = lambda _, k=key: _tile_plans.pop(k, None)   # noqa: E731
    _tile_plans[key] = (weakref.ref(rowptr, drop), weakref.ref(w, drop) if w is not None else None, plan, weakref.ref(col, drop))
    return plan


def csr_aggregate_form(rowptr, col, w, x, block_rows=0):
    """Which form of the aggregate this (graph, operand) takes — the ONE place the policy lives (this ctypes binding and the C++
    operators' caller custom_ops.csr_aggregate both ask it): ("tiled", TilePlan) where the graph is block-local with ``block_rows``
    rows per block, its plan is valid (rows in source-tile order) and the launch has enough (block, destination tile, slice)
    workgroups to fill the chip (``PREFER_TILED_AGGREGATE``); ("blocks", row order or None) when a block fits the LDS with
    16-channel slices (``PREFER_LDS_AGGREGATE``); else ("gather", None).  All three give the same bits."""
    n = rowptr.numel() - 1
    C = x.shape[1]
    if PREFER_TILED_AGGREGATE is not False and block_rows > 0 and C % 16 == 0 and n > 0 and x.data_ptr() % 16 == 0 and \
            (PREFER_TILED_AGGREGATE or -(-n // block_rows) * -(-block_rows // 2560) * (C // 16) >= TILED_MIN_WORKGROUPS):
        plan = csr_tile_plan(rowptr, col, w, block_rows)
        if plan is not None and plan.valid:
            return "tiled", plan
    rows_max = LDS_BLOCK_ROWS_MAX if PREFER_LDS_AGGREGATE else LDS_SLICE16_ROWS_MAX
    if PREFER_LDS_AGGREGATE is not False and 0 < block_rows <= rows_max and C % 4 == 0 and n > 0:
        lpr = next(c for c in (4, 2, 1) if C % (4 * c) == 0 and (block_rows + 1) * 16 * c <= 160 * 1024)
        if (lpr == 4 or PREFER_LDS_AGGREGATE) and -(-n // block_rows) * (C // (4 * lpr)) >= LDS_MIN_WORKGROUPS:
            return "blocks", (csr_block_row_order(rowptr, block_rows) if block_rows <= 16384 else None)
    return "gather", None


def csr_aggregate(rowptr, col, w, x, self_coef=None, bias=None, scale=None, shift=None, act=ACT_NONE, block_rows=0):
    """y[i] = epi(sum_e w[e] * x[col[e]] (+ (1+self_coef) * x[i]))   (gnnpn_csr_aggregate_f32).
    ``block_rows`` > 0 = the caller's promise that the graph is block-local with blocks of that many rows (graph.CSR
    records it); the form — tiled (gnnpn_csr_aggregate_tiled_f32), whole-block LDS (gnnpn_csr_aggregate_blocks_f32) or gather —
    is ``csr_aggregate_form``'s choice; bit-identical every way."""
    x = _rows2d(x, "csr_aggregate.x")
    n = rowptr.numel() - 1
    C = x.shape[1]
    form, aux = csr_aggregate_form(rowptr, col, w, x, block_rows)
    if form == "tiled":
        return aux.aggregate(x, self_coef, bias, scale, shift, act)
    y = torch.empty((n, C), dtype=F32, device=x.device)
    if form == "blocks":
        check(_lib.load().gnnpn_csr_aggregate_blocks_f32(
            dev_ptr(rowptr, I32, "rowptr"), dev_ptr(col, I32, "col"), dev_ptr(w, F32, "w", True), dev_ptr(x, F32, "x"), C,
            dev_ptr(self_coef, F32, "self_coef", True), dev_ptr(bias, F32, "bias", True),
            dev_ptr(scale, F32, "scale", True), dev_ptr(shift, F32, "shift", True), act, dev_ptr(y, F32, "y"), C, n, C,
            int(block_rows), dev_ptr(aux, I32, "row_order", True), stream_ptr()), "gnnpn_csr_aggregate_blocks_f32")
        return y
    check(_lib.load().gnnpn_csr_aggregate_f32(
        dev_ptr(rowptr, I32, "rowptr"), dev_ptr(col, I32, "col"), dev_ptr(w, F32, "w", True), dev_ptr(x, F32, "x"), C,
        dev_ptr(self_coef, F32, "self_coef", True), dev_ptr(bias, F32, "bias", True),
        dev_ptr(scale, F32, "scale", True), dev_ptr(shift, F32, "shift", True), act, dev_ptr(y, F32, "y"), C, n, C,
        stream_ptr()), "gnnpn_csr_aggregate_f32")
    return y


def gcn_norm(rowptr, col, w_raw):
    """Symmetric GCN normalisation on a self-loop-complete destination-major CSR -> norm[e]."""
    n = rowptr.numel() - 1
    dis = torch.empty(n, dtype=F32, device=w_raw.device)
    norm = torch.empty_like(w_raw)
    check(_lib.load().gnnpn_gcn_norm_f32(dev_ptr(rowptr, I32, "rowptr"), dev_ptr(col, I32, "col"),
                                         dev_ptr(w_raw, F32, "w_raw"), dev_ptr(dis, F32, "dis"),
                                         dev_ptr(norm, F32, "norm"), n, stream_ptr()), "gnnpn_gcn_norm_f32")
    return norm


def segment_mean(segptr, x):
    x = _rows2d(x, "segment_mean.x")
    n_seg, C = segptr.numel() - 1, x.shape[1]
    out = torch.empty((n_seg, C), dtype=F32, device=x.device)
    check(_lib.load().gnnpn_segment_mean_f32(dev_ptr(segptr, I32, "segptr"), dev_ptr(x, F32, "x"), C,
                                             dev_ptr(out, F32, "out"), C, n_seg, C, stream_ptr()),
          "gnnpn_segment_mean_f32")
    return out


def pack_mfma_b(w):
    """[N, K] weight (rows = output features) -> v_mfma_f32_16x16x4_f32 B-fragments [N/16, K16, 64, 4]:
    packed[t][k16][lane][j] = w[16t + lane%16][16*k16 + 4j + lane//16], K zero-padded to a multiple of 16.
    A one-time layout change at weight-load time (no arithmetic)."""
    N, K = w.shape
    if N % 16:
        raise GnnpnError(f"pack_mfma_b: output features {N} must be a multiple of 16")
    Kp = (K + 15) // 16 * 16
    wp = torch.zeros((N, Kp), dtype=w.dtype, device=w.device)
    wp[:, :K] = w
    v = wp.view(N // 16, 16, Kp // 16, 4, 4)              # [t, c, k16, j, kq]
    return v.permute(0, 2, 4, 1, 3).reshape(N // 16, Kp // 16, 64, 4).contiguous()   # [t, k16, (kq, c), j]


def gin_layer(rowptr, col, x, eps, w1, b1, a1, s1, w2, b2, a2, s2, w3=None, b3=None):
    """One GIN layer for large workflow graphs in one launch (gnnpn_gin_layer_f32): aggregate -> Linear + BN + ReLU -> Linear + BN
    + ReLU [-> Linear + bias].  w1 / w2 / w3: the weights PACKED by ``pack_mfma_b32`` ([N/32, Kp/2, 64]).  Bit-identical to
    csr_aggregate(self_coef=eps) + linear + linear (+ linear) on the unpacked weights.  Raises GnnpnError for shapes the kernel
    is not built for (``gin_layer_supported``)."""
    x = _rows2d(x, "gin_layer.x")
    n, c_in = x.shape
    if w1.dim() != 3 or w2.dim() != 3 or (w3 is not None and w3.dim() != 3) or w1.shape[1] * 2 != (c_in + 31) // 32 * 32:
        raise GnnpnError("gin_layer: weights must be packed with ops.pack_mfma_b32 (and w1 for this many input channels)")
    h1, h2 = w1.shape[0] * 32, w2.shape[0] * 32
    h3 = w3.shape[0] * 32 if w3 is not None else 0
    out = torch.empty((n, h3 if w3 is not None else h2), dtype=F32, device=x.device)
    check(_lib.load().gnnpn_gin_layer_f32(
        dev_ptr(rowptr, I32, "rowptr"), dev_ptr(col, I32, "col"), dev_ptr(x, F32, "x"), c_in, c_in, dev_ptr(eps, F32, "eps"),
        dev_ptr(w1, F32, "w1"), dev_ptr(b1, F32, "b1", True), dev_ptr(a1, F32, "a1", True), dev_ptr(s1, F32, "s1", True), h1,
        dev_ptr(w2, F32, "w2"), dev_ptr(b2, F32, "b2", True), dev_ptr(a2, F32, "a2", True), dev_ptr(s2, F32, "s2", True), h2,
        dev_ptr(w3, F32, "w3", True), dev_ptr(b3, F32, "b3", True), h3, dev_ptr(out, F32, "out"), out.shape[1], n, stream_ptr()),
        "gnnpn_gin_layer_f32")
    return out


def pack_split_weights(w):
    """[N, K] fp32 weight (rows = output features, N a multiple of 16) -> (packed uint8 image, col_inv [N]) for ``gin_layer_split``:
    every element decomposed exactly into three fp16 pieces under a per-column power-of-two scale, in the fp16 matrix core's
    B-fragment order (gnnpn_pack_split_weights_f16).  Once per model."""
    w = _rows2d(w, "pack_split_weights.w")
    N, K = w.shape
    nbytes = _lib.load().gnnpn_split_weights_bytes(N, K)
    if nbytes <= 0:
        raise GnnpnError(f"pack_split_weights: output features {N} must be a multiple of 16")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    inv = torch.empty(N, dtype=F32, device=w.device)
    check(_lib.load().gnnpn_pack_split_weights_f16(dev_ptr(w, F32, "w"), K, N, K, dev_ptr(packed, torch.uint8, "packed"), dev_ptr(inv, F32, "col_inv"),
                                                   stream_ptr()), "gnnpn_pack_split_weights_f16")
    return packed, inv


def gin_layer_split(rowptr, col, x, eps, w1, i1, b1, a1, s1, w2, i2, b2, a2, s2, w3=None, i3=None, b3=None):
    """``gin_layer`` with the three dense products on the fp16 matrix cores through the exact split (gnnpn_gin_layer_split):
    (w, i) pairs from ``pack_split_weights``.  Built for c_in -> 256 -> 128 (-> 128)."""
    x = _rows2d(x, "gin_layer_split.x")
    n, c_in = x.shape
    lib = _lib.load()
    k1 = (c_in + 31) // 32 * 32
    if w1.dtype != torch.uint8 or w2.dtype != torch.uint8 or (w3 is not None and w3.dtype != torch.uint8) or i1 is None or i2 is None:
        raise GnnpnError("gin_layer_split: weights must be packed with ops.pack_split_weights")
    h1, h2 = i1.numel(), i2.numel()
    h3 = i3.numel() if w3 is not None else 0
    if (w1.numel() != lib.gnnpn_split_weights_bytes(h1, k1) or w2.numel() != lib.gnnpn_split_weights_bytes(h2, h1) or
            (w3 is not None and w3.numel() != lib.gnnpn_split_weights_bytes(h3, h2))):
        raise GnnpnError("gin_layer_split: packed weights do not chain (c_in -> h1 -> h2 [-> h3])")
    out = torch.empty((n, h3 if w3 is not None else h2), dtype=F32, device=x.device)
    u8 = lambda t, name, opt=False: None if (t is None and opt) else dev_ptr(t, torch.uint8, name)   # noqa: E731
    check(lib.gnnpn_gin_layer_split(
        dev_ptr(rowptr, I32, "rowptr"), dev_ptr(col, I32, "col"), dev_ptr(x, F32, "x"), c_in, c_in, dev_ptr(eps, F32, "eps"),
        u8(w1, "w1"), dev_ptr(i1, F32, "i1"), dev_ptr(b1, F32, "b1", True), dev_ptr(a1, F32, "a1", True), dev_ptr(s1, F32, "s1", True), h1,
        u8(w2, "w2"), dev_ptr(i2, F32, "i2"), dev_ptr(b2, F32, "b2", True), dev_ptr(a2, F32, "a2", True), dev_ptr(s2, F32, "s2", True), h2,
        u8(w3, "w3", True), dev_ptr(i3, F32, "i3", True), dev_ptr(b3, F32, "b3", True), h3, dev_ptr(out, F32, "out"), out.shape[1], n,
        stream_ptr()), "gnnpn_gin_layer_split")
    return out


def gin_layer_split_supported(c_in, h1, h2, h3=None):
    return h1 == 256 and h2 == 128 and (h3 is None or h3 == 128) and 0 < c_in <= 128 and (c_in % 4 == 0 or c_in <= 32)


def gin_layer_supported(c_in, h1, h2, h3=None):
    return h1 == 256 and h2 == 128 and (h3 is None or h3 == 128) and 0 < c_in <= 256


def pack_mfma_b32(w):
    """[N, K] weight (rows = output features) -> v_mfma_f32_32x32x2_f32 B-fragments [N/32, Kp/2, 64]:
    packed[t][kp][lane] = w[32t + lane%32][2*kp + lane//32], K zero-padded to a multiple of 32 (the padding linear_f32_kernel
    applies in its last k-tile).  A one-time layout change at weight-load time (no arithmetic)."""
    N, K = w.shape
    if N % 32:
        raise GnnpnError(f"pack_mfma_b32: output features {N} must be a multiple of 32")
    Kp = (K + 31) // 32 * 32
    wp = torch.zeros((N, Kp), dtype=w.dtype, device=w.device)
    wp[:, :K] = w
    v = wp.view(N // 32, 32, Kp // 2, 2)                  # [t, c, kp, h]
    return v.permute(0, 2, 3, 1).reshape(N // 32, Kp // 2, 64).contiguous()   # [t, kp, (h, c)]


REQUEST_BRANCH_MAX_NODES = 16


def request_branch(x, table, rowptr, col, seg_ptr, max_nodes, layers, lin_w_packed, lin_b, hidden):
    """The whole GIN branch in one launch (gnnpn_request_branch_f32): x [N,1+f] node rows, CSR of the batched workflow
    graphs (every edge inside its graph, at most 16 nodes per graph) -> [B, hidden].  ``layers``: list of dicts with
    w0p, b0, a1, s1, w3p, b3, a2, s2 (packed weights: pack_mfma_b; BN folded) and eps.  Raises GnnpnError for shapes
    the fused kernel is not built for (callers fall back to the separate kernels)."""
    x = _rows2d(x, "request_branch.x")
    vocab, emb = table.shape
    B = seg_ptr.numel() - 1
    out = torch.empty((B, hidden), dtype=F32, device=x.device)
    arr = (_lib.GinLayer * len(layers))()
    for i, lp in enumerate(layers):
        a = arr[i]
        for field, key in (("w0_packed", "w0p"), ("b0", "b0"), ("bn1_scale", "a1"), ("bn1_shift", "s1"), ("w3_packed", "w3p"),
                           ("b3", "b3"), ("bn2_scale", "a2"), ("bn2_shift", "s2"), ("eps", "eps")):
            setattr(a, field, dev_ptr(lp[key], F32, f"layers[{i}].{key}").value)
    check(_lib.load().gnnpn_request_branch_f32(
        dev_ptr(x, F32, "x"), x.shape[1] - 1, dev_ptr(table, F32, "table"), vocab, emb, dev_ptr(rowptr, I32, "rowptr"),
        dev_ptr(col, I32, "col"), dev_ptr(seg_ptr, I32, "seg_ptr"), B, int(max_nodes), len(layers), arr, int(hidden),
        dev_ptr(lin_w_packed, F32, "lin_w_packed"), dev_ptr(lin_b, F32, "lin_b"), dev_ptr(out, F32, "out"), stream_ptr()),
        "gnnpn_request_branch_f32")
    return out


def select_candidates(scores, cat_ptr, qos, local_bounds, present, global_bounds, n_per):
    """Per (problem, category) top-``n_per`` feasible services -> (rows [B,L,8] fp32, ids [B,L] int32)."""
    scores = _rows2d(scores, "select.scores")
    B, S = scores.shape
    T = cat_ptr.numel() - 1
    if qos.shape != (S, 4) or local_bounds.shape != (B, T, 4) or present.shape != (B, T) \
            or global_bounds.shape != (B, 4):
        raise GnnpnError("select_candidates: inconsistent shapes")
    rows = torch.empty((B, T * n_per, 8), dtype=F32, device=scores.device)
    ids = torch.empty((B, T * n_per), dtype=I32, device=scores.device)
    check(_lib.load().gnnpn_select_candidates(
        dev_ptr(scores, F32, "scores"), S, dev_ptr(cat_ptr, I32, "cat_ptr"), dev_ptr(qos, F64, "qos"),
        dev_ptr(local_bounds, F64, "local_bounds"), dev_ptr(present, U8, "present"),
        dev_ptr(global_bounds, F64, "global_bounds"), dev_ptr(rows, F32, "rows"), dev_ptr(ids, I32, "ids"), B, T,
        n_per, stream_ptr()), "gnnpn_select_candidates")
    return rows, ids


def rank_rows(scores):
    """Full descending ranking per row, ties -> lowest id (int32 [B,S])."""
    scores = _rows2d(scores, "rank_rows.scores")
    B, S = scores.shape
    ranking = torch.empty((B, S), dtype=I32, device=scores.device)
    check(_lib.load().gnnpn_rank_rows(dev_ptr(scores, F32, "scores"), S, dev_ptr(ranking, I32, "ranking"), B, S,
                                      stream_ptr()), "gnnpn_rank_rows")
    return ranking


def precision_at_k(ranking, labels, ks=(1, 5)):
    """[B, n_k] fraction of the top-k ranked services with label 1 (trainML.py:63-70)."""
    B, S = labels.shape
    kt = torch.tensor(list(ks), dtype=I32, device=labels.device)
    out = torch.empty((B, len(ks)), dtype=F32, device=labels.device)
    check(_lib.load().gnnpn_precision_at_k(dev_ptr(ranking, I32, "ranking"), ranking.shape[1],
                                           dev_ptr(labels, F32, "labels"), S, B, S, dev_ptr(kt, I32, "ks"), len(ks),
                                           dev_ptr(out, F32, "out"), stream_ptr()), "gnnpn_precision_at_k")
    return out


def pack_lstm_weight(w):
    """[4H, H] (gate-major rows, torch.nn.LSTM layout) -> [H/4, 4, H, 4]: element [k4][g][j][i] =
    w[g*H + j][4*k4 + i].  A one-time layout change at weight-load time (no arithmetic)."""
    H = w.shape[1]
    if w.shape[0] != 4 * H or H % 4:
        raise GnnpnError(f"pack_lstm_weight: expected [4H,H], got {tuple(w.shape)}")
    return w.reshape(4, H, H // 4, 4).permute(2, 0, 1, 3).contiguous()


def pack_lstm_split_weights(whh_packed):
    """The exact split ("split" precision) of a packed recurrent weight matrix [H/4, 4, H, 4], H = 256, made once per model
    (gnnpn_lstm_pack_split_weights_f32): what every cooperative launch otherwise works out for itself from ``whh_packed`` —
    same bits, 9-13 us less per launch.  -> uint8 buffer for the ``whh_split`` entry of lstm_encode's / pointer_decode's nets."""
    if whh_packed.numel() != 4 * 256 * 256:
        raise GnnpnError(f"pack_lstm_split_weights: a packed [64, 4, 256, 4] matrix (H = 256) expected, got {tuple(whh_packed.shape)}")
    lib = _lib.load()
    out = torch.empty(int(lib.gnnpn_lstm_split_weights_bytes()), dtype=U8, device=whh_packed.device)
    check(lib.gnnpn_lstm_pack_split_weights_f32(dev_ptr(whh_packed.contiguous(), F32, "whh_packed"), dev_ptr(out, U8, "split"), stream_ptr()),
          "gnnpn_lstm_pack_split_weights_f32")
    return out


# operand precision of the recurrent W_hh.h product -> GNNPN_PREC_* (include/gnnpn_hip.h)
_PRECISIONS = {"f32": 0, "f16": 1, "split": 2}


class Workspaces:
    """Hand-off workspaces + the sticky status word of the cooperative recurrent kernels, for ONE stream of launches.

    Launches that may be in flight at the same time (two pipelined steps on two streams) must not share hand-off
    buffers: each gets its own ``Workspaces`` (``PipelinedRunner`` owns one per slot).  A buffer is never freed or
    replaced once handed out — a captured HIP graph may have baked its address in — so growing keeps the old tensor
    alive; while ``frozen`` (set by whoever captured a graph over it) growing raises instead.
    ``status`` is gnnpn_launch_opts_t.sticky_status (include/gnnpn_hip.h, GNNPN_STATUS_*): word 0 — every cooperative launch ORs
    its failure code (a bounded inter-workgroup wait timed out: outputs invalid) into it and nothing but ``check()`` / ``poll()``
    clears it, so one host read covers every launch since the last check; words 4..7 — the proof of work: workgroup-tiles the
    launches were expected to finish and workgroup-tiles their workgroups did finish (encoder, decoder), which must be equal after
    a synchronisation.  ``launched`` is the HOST's own count of the same quantity (what the wrappers and the graph replays of this
    process asked for): ``check`` / ``poll`` compare all three and report a difference as code 16 (GNNPN_COOP_SHORTFALL) — a
    launch that did nothing raises no code by itself."""

    WORDS = 8                             # GNNPN_STATUS_WORDS
    SHORTFALL = 16                        # GNNPN_COOP_SHORTFALL

    def __init__(self, device):
        self.device = torch.device(device)
        self.id = -1                      # index in the process-wide registry (custom_ops pass it as an int)
        self.status = torch.zeros(self.WORDS, dtype=torch.int32, device=self.device)
        self.launched = [0, 0]            # workgroup-tiles asked of the encoder / decoder launches on this object (host count)
        self._captured = [0, 0]           # ... of the launches recorded into a HIP graph that nobody has claimed yet (take_captured)
        self._untracked = False           # a captured graph replays these launches without telling: the host count is unknown
        self._warned_untracked = False    # ... said once per object (_read)
        self.last_progress = None         # the counters the last poll / check read (diagnosis; bench.py's per_rank)
        self.last_seats = None            # ... and the cumulative placement events (declined / off-canonical seats)
        self._encode = None
        self._decode = None
        self._retired = []
        self.frozen = False

    def encode(self):
        if self._encode is None:
            n = int(_lib.load().gnnpn_lstm_encode_workspace_bytes())
            self._encode = torch.zeros(n, dtype=torch.uint8, device=self.device)
        return self._encode

    def decode(self, B, T, n_per):
        need = int(_lib.load().gnnpn_pointer_decode_workspace_bytes(B, T, n_per))
        ws = self._decode
        if ws is None or ws.numel() < need:
            if ws is not None:
                if self.frozen:
                    raise GnnpnError(f"decode workspace of {ws.numel()} B is captured in a HIP graph and cannot grow to "
                                     f"{need} B: use a separate Workspaces for the larger shape")
                self._retired.append(ws)
            ws = self._decode = torch.zeros(need, dtype=torch.uint8, device=self.device)
        return ws

    @staticmethod
    def coop_units(n_nets, n_problems):
        """Workgroup-tiles of one cooperative launch: 8 members x nets x tiles of 16 problems (what the launch books as expected)."""
        return 8 * int(n_nets) * ((int(n_problems) + 15) // 16)

    def note_launch(self, which, units):
        """Host-side bookkeeping of one cooperative launch (``which``: 0 encoder, 1 decoder) — called by the wrappers that make
        the launch and, per replay, by whoever captured them into a HIP graph."""
        if torch.cuda.is_current_stream_capturing():
            # recorded, not run: whoever captures claims the tally (take_captured) and books it per replay (graph_replay);
            # until then the replays are invisible to the host count, which _read then leaves out of the comparison
            self._captured[which] += int(units)
            self._untracked = True
            return
        self.launched[which] = (self.launched[which] + int(units)) & 0xFFFFFFFF

    def take_captured(self):
        """[encoder, decoder] workgroup-tiles of the launches recorded on this object since the last call — what ONE replay of
        the graph just captured will ask for (ops.graph_replay books it per replay)."""
        out, self._captured, self._untracked = self._captured, [0, 0], False
        return out

    def _read(self):
        """Synchronise, read the status block once.  -> (code incl. SHORTFALL, progress dict)"""
        torch.cuda.synchronize(self.device)
        w = [v & 0xFFFFFFFF for v in self.status.tolist()]
        prog = {"encoder": {"expected": w[4], "finished": w[5], "host_expected": self.launched[0]},
                "decoder": {"expected": w[6], "finished": w[7], "host_expected": self.launched[1]}}
        word = w[0]
        if any(p["finished"] != p["expected"] or (not self._untracked and p["expected"] != p["host_expected"]) for p in prog.values()):
            word |= self.SHORTFALL
        # placement events summed over every launch on this object since the status block was last cleared (GNNPN_STATUS_DECLINED_SEATS /
        # _OFF_CANONICAL_SEATS, ABI 9): a slow multi-GPU line can then be read from the JSON alone (bench.py's per_rank)
        self.last_seats = {"declined": w[1], "off_canonical": w[2]}
        if self._untracked and not self._warned_untracked:
            # a capture made outside ops.graph_replay replays cooperative launches without booking them: finished == expected is
            # still checked, expected == the host's own count is not (and stays off for this object) — say so, once
            import warnings
            warnings.warn("gnnpn: cooperative launches on this Workspaces were captured into a HIP graph that ops.graph_replay does "
                          "not wrap; the host-count half of the proof of work (expected == what the host asked for) is off for it — "
                          "wrap the graph with ops.graph_replay(graph, [workspaces]) to keep it", RuntimeWarning, stacklevel=3)
            self._warned_untracked = True
        self.last_progress = prog
        return word, prog

    def _clear(self):
        self.status.zero_()
        self.launched = [0, 0]
        with torch.cuda.device(self.device):
            return _lib.load().gnnpn_coop_reset_staffing()

    def poll(self):
        """Synchronise the device, return the status code and clear it (0: every launch since the last poll / check raised
        nothing AND finished every workgroup-tile it was expected to) — the non-raising form of ``check`` for callers that have a
        degraded mode to fall back to (PipelinedRunner, bench.py)."""
        word, _ = self._read()
        if word != 0:
            self._clear()
        return word

    def placement(self):
        """Placement counters of this object's LAST encoder launch (the launch's own status area, zeroed before every launch):
        members placed, seats taken off their canonical CU by the reserve, early arrivals that declined a seat because of their
        LDS position (csrc/coop_common.h) — diagnosis only; synchronises."""
        torch.cuda.synchronize(self.device)
        if self._encode is None:
            return None
        w = self._encode[:16].view(torch.int32).tolist()
        placed = self._encode[14336:15360].view(torch.int32)[::32].sum().item()     # per-XCD words on lines of their own (coop_common.h: COOP_PLACED_OFFSET)
        return {"members_placed": int(placed), "off_canonical_seats": w[2], "declined_seats": w[3]}

    def check(self, what="cooperative kernel"):
        """Synchronise the device and raise if any launch since the last check reported a failed hand-off."""
        word, prog = self._read()
        if word != 0:
            # a launch that timed out leaves the count of staffing launches by itself (coop_raise); this reset is the belt to
            # those braces — on THIS object's device, and a failure of the reset is reported with the status it hides behind
            rc = self._clear()
            if rc != 0:
                what = f"{what} (and gnnpn_coop_reset_staffing failed: {_lib.load().gnnpn_last_error().decode('utf-8', 'replace')})"
            last = [int(b[:4].view(torch.int32).item()) if b is not None else None for b in (self._encode, self._decode)]
            detail = ""
            if word & 2:
                rec = decode_failure_record()
                detail = f"; decoder failure record: {rec['failures']} sweeps, first: {rec['records'][:4]}"
            err = GnnpnError(f"{what}: status {word:#x} — at least one cooperative launch since the last check failed (its outputs "
                             f"are invalid); bits: 1 encoder sweep timed out, 2 decoder sweep timed out, 4 a group member never "
                             f"showed up, 8 the workspace was not clean when a launch began, 16 workgroup-tiles finished != expected "
                             f"(a launch did not do its work): {prog}; last launches' own words (encoder, decoder) = {last}{detail}")
            err.status = word                 # the word itself, for callers with a degraded mode (PipelinedRunner.synchronize): no parsing of the text
            raise err


class lds_footprint:
    """``with ops.lds_footprint(kb):`` — the ordinary LDS-using kernels this thread launches inside the block (gnnpn_linear_f32,
    gnnpn_request_branch_f32) are padded to ``kb`` KB of LDS per workgroup (gnnpn_lds_footprint_kb): PipelinedRunner captures
    its free-running slots' graphs that way, so that a small kernel's LDS range is exactly what the other slot's cooperative
    workgroup needs when it is freed (include/gnnpn_hip.h)."""

    def __init__(self, kb):
        self.kb, self.prev = int(kb), 0

    def __enter__(self):
        self.prev = _lib.load().gnnpn_lds_footprint_kb(self.kb)
        return self

    def __exit__(self, *exc):
        _lib.load().gnnpn_lds_footprint_kb(self.prev)
        return False


def graph_replay(graph, all_ws):
    """Call right after capturing ``graph`` (torch.cuda.CUDAGraph) over launches on the ``Workspaces`` in ``all_ws``: returns
    ``replay()`` = ``graph.replay()`` + the host-side booking of the cooperative launches one replay makes (proof of work:
    ``Workspaces.launched`` against the device's expected / finished counters)."""
    per = [(w, w.take_captured()) for w in all_ws]

    def replay():
        graph.replay()
        for w, (enc, dec) in per:
            w.note_launch(0, enc)
            w.note_launch(1, dec)
    return replay


def decode_failure_record(clear=True):
    """gnnpn_decode_diag as a list of dicts (one per timed-out decoder sweep, the first 31) — diagnosis only."""
    import ctypes
    buf = (ctypes.c_uint32 * 512)()
    _lib.check(_lib.load().gnnpn_decode_diag(ctypes.cast(buf, ctypes.c_void_p), 512, 1 if clear else 0), "gnnpn_decode_diag")
    names = ("group", "member", "tile", "k", "wave", "tag", "h_missing_members", "p_missing_members", "latent_missing",
             "claims_xcd0_3", "claims_xcd4_7", "t_lo", "t_hi", "gpx", "workgroup", "launch_status")
    n = int(buf[0])
    return {"failures": n, "records": [dict(zip(names, [int(v) for v in buf[16 * (i + 1):16 * (i + 2)]])) for i in range(min(n, 31))]}


_default_ws = {}
# id -> Workspaces, by WEAK reference: a PipelinedRunner / capture that is dropped takes its workspaces with
# it, and check_status() only visits the ones still alive (it synchronises and reads one word per registered object)
_all_ws = weakref.WeakValueDictionary()
_ws_ids = itertools.count()


def new_workspaces(device):
    w = Workspaces(device)
    w.id = next(_ws_ids)
    _all_ws[w.id] = w
    return w


def workspaces_by_id(ws_id):
    w = _all_ws.get(ws_id)
    if w is None:
        raise GnnpnError(f"workspaces #{ws_id} no longer exist (their owner was dropped)")
    return w


def workspaces(device, ws=None):
    """``ws`` itself, or the process-wide default ``Workspaces`` of ``device`` (single-stream use)."""
    if ws is not None:
        return ws
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _default_ws:
        _default_ws[key] = new_workspaces(torch.device(*key))
    return _default_ws[key]


def set_option(name, value):
    """Diagnostics switch of the library (gnnpn_set_option): only "lstm_ablate" (tools/).  Implementation choice,
    placement and hand-off form are per-call arguments (``impl``, ``lds_kb``, ``write_through``)."""
    check(_lib.load().gnnpn_set_option(name.encode(), int(value)), "gnnpn_set_option")


def check_status(device=None):
    """Synchronise and raise if a bounded inter-workgroup wait of ANY cooperative launch since the last check timed out
    (the sticky status word of every Workspaces of this process, optionally of one device only)."""
    want = None if device is None else torch.device(device)
    for w in list(_all_ws.values()):
        if want is None or w.device.type == want.type and (want.index is None or w.device.index == want.index):
            w.check()


def run_checked(fn, device=None, retries=1):
    """``fn(attempt)`` followed by ``check_status(device)``; if a cooperative launch of that attempt reported a failed
    inter-workgroup hand-off (its outputs are invalid) the batch is run AGAIN — ``attempt`` 1, 2, .. — up to ``retries`` times
    before the error is raised.  The drivers that write artefacts (ML2PN.infer, evalPN.evaluate) call their batches through this
    and pass ``write_through = attempt > 0`` (the placement-independent hand-off form) to the repeat: a time-out is rare and
    box-dependent (profiles/LOG_r01_r04.md section 13.3), results never silently come from a failed launch, and one bad launch does not end a
    run over thousands of batches.  Every repeat is reported with ``warnings.warn``."""
    import warnings
    for attempt in range(retries + 1):
        out = fn(attempt)
        try:
            check_status(device)
            return out
        except GnnpnError as e:
            if attempt == retries:
                raise
            warnings.warn(f"gnnpn: a cooperative launch reported a failed hand-off; running the batch again "
                          f"(attempt {attempt + 2} of {retries + 1}, write-through hand-off): {str(e)[:200]}", RuntimeWarning)


def _launch_opts(ws, impl, lds_kb, write_through, paired_start=False):
    o = _lib.LaunchOpts()
    o.impl, o.lds_kb, o.write_through = int(impl), int(lds_kb), int(bool(write_through))
    o.paired_start = int(bool(paired_start))
    o.sticky_status = ws.status.data_ptr() if ws is not None else None
    return o


def coop_supported(H, n_per=1, impl=0):
    """Shapes the cooperative recurrent kernels are built for (else: per-workgroup streaming);
    ``impl`` 1 forces the streaming form."""
    return H == 256 and n_per <= 16 and impl != 1


def lstm_encode(nets, precision="f32", impl=0, lds_kb=0, write_through=False, ws=None, paired_start=False):
    """Run the encoder recurrence of len(nets) nets in ONE launch (gnnpn_lstm_encode_f32).

    precision="f16" (opt-in, cooperative form only): W_hh and h_{t-1} enter the recurrent product as
    fp16 with fp32 accumulation (BASELINE configs[4] "fp16 encoder MFMA path"); everything else stays
    fp32.  Not parity-exact — report an agreement rate against "f32".

    nets: list of dicts with whh (packed), bhh and EITHER pregates [B,L,4H] OR inputs [B,L,8] +
    w_in [4H,8] + b_in [4H] (input projection evaluated inside the cooperative kernel; for shapes
    without a cooperative kernel the projection is materialised first with gnnpn_linear_f32 —
    the same k-ordered fma chain + bias, so the same bits).
    impl: 0 auto, 1 per-workgroup streaming, 2 cooperative; lds_kb / write_through / paired_start (this launch starts together
    with a partner launch of the same LDS footprint: strict placement, include/gnnpn_hip.h) / ws: gnnpn_launch_opts_t and the
    ``Workspaces`` to use (default: the device's shared one).
    -> (enc_out list [B,L,H], h_n list [B,H], c_n list [B,H])."""
    n = len(nets)
    H = nets[0]["bhh"].numel() // 4
    first = nets[0]["pregates"] if nets[0].get("pregates") is not None else nets[0]["inputs"]
    B, L = first.shape[0], first.shape[1]
    dev = first.device
    coop = coop_supported(H, impl=impl)
    arr = (_lib.EncodeNet * n)()
    enc, h_n, c_n, keep = [], [], [], []
    for i, d in enumerate(nets):
        pre = d.get("pregates")
        if pre is None and not coop:
            x = d["inputs"]
            pre = linear(x.reshape(B * L, x.shape[2]), d["w_in"], d["b_in"]).view(B, L, 4 * H)
            keep.append(pre)
        e = torch.empty((B, L, H), dtype=F32, device=dev)
        hn = torch.empty((B, H), dtype=F32, device=dev)
        cn = torch.empty((B, H), dtype=F32, device=dev)
        enc.append(e), h_n.append(hn), c_n.append(cn)
        a = arr[i]
        a.pregates = None if pre is None else dev_ptr(pre, F32, f"nets[{i}].pregates").value
        if pre is None:
            a.inputs = dev_ptr(d["inputs"], F32, f"nets[{i}].inputs").value
            a.w_in = dev_ptr(d["w_in"], F32, f"nets[{i}].w_in").value
            a.b_in = dev_ptr(d["b_in"], F32, f"nets[{i}].b_in").value
        a.whh_packed = dev_ptr(d["whh"], F32, f"nets[{i}].whh").value
        a.whh_split = dev_ptr(d.get("whh_split"), U8, f"nets[{i}].whh_split", True).value if d.get("whh_split") is not None else None
        a.bhh = dev_ptr(d["bhh"], F32, f"nets[{i}].bhh").value
        a.enc_out, a.h_n, a.c_n = (dev_ptr(t, F32, "out").value for t in (e, hn, cn))
    wsp = workspaces(dev, ws) if coop else None
    buf = wsp.encode() if coop else None
    if precision not in _PRECISIONS:
        raise GnnpnError(f"lstm_encode: unknown precision {precision!r}")
    if precision != "f32" and not coop:
        raise GnnpnError(f"lstm_encode: precision={precision!r} needs the cooperative form (H = 256)")
    opts = _launch_opts(wsp, impl, lds_kb, write_through, paired_start)
    check(_lib.load().gnnpn_lstm_encode_f32(n, arr, B, L, H, 8, _PRECISIONS[precision], _lib.ctypes.byref(opts),
                                            dev_ptr(buf, torch.uint8, "workspace", True),
                                            0 if buf is None else buf.numel(), stream_ptr()), "gnnpn_lstm_encode_f32")
    if coop:
        wsp.note_launch(0, _lib.load().gnnpn_last_launch_units())
    return enc, h_n, c_n


def pointer_decode(nets, inputs, n_cat, n_per, tanh_c=10.0, use_tanh=True, want_queries=False, precision="f32",
                   impl=0, lds_kb=0, write_through=False, ws=None, paired_start=False):
    """Greedy decode of 1 or 2 pointer networks in ONE call (gnnpn_pointer_decode_f32).

    nets: list of dicts with keys enc_out, h0, c0, start, wih, whh, bih, bhh, EITHER embedded [B,L,H]
    OR emb_w [H,8] + emb_b [H] (picked rows embedded in-kernel), and optionally latent_win ([B,T,K]
    tensor computed earlier) or latent_from (index of an earlier net of this call), and optionally sample=True +
    sample_seed (the pick of every step is drawn from the window softmax: gnnpn_decode_net_t.sample).
    Returns one dict per net: idx [B,T] i32, win_logits [B,T,K], pick_prob [B,T], actions [B,T,8],
    queries [B,T,H] | None.
    precision="split": the W_hh.h product from exact three-piece fp16 operands (cooperative, folded form only); "f16" is an
    encoder-only mode and leaves the decoder in fp32.
    impl: 0 auto, 1 streaming, 2 cooperative (8-CU groups), 3 (16-CU groups), 4 (8-CU groups, 256-register build for two
    workgroups per CU); lds_kb / write_through / ws as for lstm_encode."""
    B, L, H = nets[0]["enc_out"].shape
    if precision not in _PRECISIONS:
        raise GnnpnError(f"pointer_decode: unknown precision {precision!r}")
    if L != n_cat * n_per:
        raise GnnpnError(f"pointer_decode: seq_len {L} != {n_cat}*{n_per}")   # modelPN.py:182
    dev = nets[0]["enc_out"].device
    arr = (_lib.DecodeNet * len(nets))()
    outs = []
    for i, d in enumerate(nets):
        out = {"idx": torch.empty((B, n_cat), dtype=I32, device=dev),
               "win_logits": torch.empty((B, n_cat, n_per), dtype=F32, device=dev),
               "pick_prob": torch.empty((B, n_cat), dtype=F32, device=dev),
               "actions": torch.empty((B, n_cat, 8), dtype=F32, device=dev),
               "queries": torch.empty((B, n_cat, H), dtype=F32, device=dev) if want_queries else None}
        outs.append(out)
        a = arr[i]
        for name, key in (("enc_out", "enc_out"), ("h0", "h0"), ("c0", "c0"), ("start", "start"),
                          ("wih_packed", "wih"), ("whh_packed", "whh"), ("bih", "bih"), ("bhh", "bhh")):
            setattr(a, name, dev_ptr(d[key], F32, f"nets[{i}].{key}").value)
        if d.get("whh_split") is not None:
            a.whh_split = dev_ptr(d["whh_split"], U8, f"nets[{i}].whh_split").value
        coop = coop_supported(H, n_per, impl)
        if d.get("xw_fold") is not None and coop:                          # folded input side (cooperative form)
            a.xw_fold = dev_ptr(d["xw_fold"], F32, f"nets[{i}].xw_fold").value
            a.xb_fold = dev_ptr(d["xb_fold"], F32, f"nets[{i}].xb_fold").value
            a.start_fold = dev_ptr(d["start_fold"], F32, f"nets[{i}].start_fold").value
        emb = d.get("embedded")
        if emb is None and not coop:                                       # no in-kernel embedding there
            emb = linear(inputs.reshape(B * L, inputs.shape[2]), d["emb_w"], d["emb_b"]).view(B, L, H)
            outs[-1]["_embedded"] = emb
        a.embedded = None if emb is None else dev_ptr(emb, F32, f"nets[{i}].embedded").value
        if emb is None:
            a.emb_w = dev_ptr(d["emb_w"], F32, f"nets[{i}].emb_w").value
            a.emb_b = dev_ptr(d["emb_b"], F32, f"nets[{i}].emb_b").value
        lw = d.get("latent_win")
        a.latent_win = None if lw is None else dev_ptr(lw, F32, f"nets[{i}].latent_win").value
        a.latent_from = int(d.get("latent_from", -1))
        a.sample = int(bool(d.get("sample", False)))
        a.sample_seed = int(d.get("sample_seed", 0)) & 0xFFFFFFFFFFFFFFFF
        a.idx = dev_ptr(out["idx"], I32, "idx").value
        a.win_logits = dev_ptr(out["win_logits"], F32, "win").value
        a.pick_prob = dev_ptr(out["pick_prob"], F32, "prob").value
        a.actions = dev_ptr(out["actions"], F32, "actions").value
        a.queries = None if out["queries"] is None else dev_ptr(out["queries"], F32, "queries").value
    coop = coop_supported(H, n_per, impl)
    wsp = workspaces(dev, ws) if coop else None
    buf = wsp.decode(B, n_cat, n_per) if coop else None
    opts = _launch_opts(wsp, impl, lds_kb, write_through, paired_start)
    check(_lib.load().gnnpn_pointer_decode_f32(
        len(nets), arr, dev_ptr(inputs, F32, "inputs"), float(tanh_c), int(bool(use_tanh)), B, n_cat, n_per, H,
        _PRECISIONS["split"] if precision == "split" else 0, _lib.ctypes.byref(opts),
        dev_ptr(buf, torch.uint8, "workspace", True), 0 if buf is None else buf.numel(), stream_ptr()),
        "gnnpn_pointer_decode_f32")
    if coop:
        wsp.note_launch(1, _lib.load().gnnpn_last_launch_units())
    return outs


ATTENTION_NAMES = {"Dot": 0, "Bahdanau": 1}


def pointer_decode_attn(net, inputs, n_cat, n_per, attention="Dot", n_glimpses=0, pointer=None, glimpse=None, tanh_c=10.0,
                        use_tanh=True, want_queries=False):
    """Greedy decode of ONE pointer network with 'Bahdanau' attention and / or glimpse rounds (gnnpn_pointer_decode_attn_f32;
    modelPN.py:80-122,204-239 — the forms the reference's configurations leave switched off).
    net: as one entry of pointer_decode's ``nets`` (embedded or emb_w/emb_b, enc_out, h0, c0, start, wih, whh, bih, bhh,
    latent_win).  pointer / glimpse ('Bahdanau' only): dicts with wq [H,H], bq [H], wref [H,H] or [H,H,1], bref [H], v [H];
    W_ref(enc_out) is formed here with gnnpn_linear_f32."""
    if attention not in ATTENTION_NAMES:
        raise NotImplementedError(attention)                            # modelPN.py:116-117
    B, L, H = net["enc_out"].shape
    if L != n_cat * n_per:
        raise GnnpnError(f"pointer_decode_attn: seq_len {L} != {n_cat}*{n_per}")
    dev = net["enc_out"].device
    keep = []
    emb = net.get("embedded")
    if emb is None:
        emb = linear(inputs.reshape(B * L, inputs.shape[2]), net["emb_w"], net["emb_b"]).view(B, L, H)
    out = {"idx": torch.empty((B, n_cat), dtype=I32, device=dev),
           "win_logits": torch.empty((B, n_cat, n_per), dtype=F32, device=dev),
           "pick_prob": torch.empty((B, n_cat), dtype=F32, device=dev),
           "actions": torch.empty((B, n_cat, 8), dtype=F32, device=dev),
           "queries": torch.empty((B, n_cat, H), dtype=F32, device=dev) if want_queries else None}
    a = _lib.DecodeNet()
    for name, key in (("enc_out", "enc_out"), ("h0", "h0"), ("c0", "c0"), ("start", "start"), ("wih_packed", "wih"),
                      ("whh_packed", "whh"), ("bih", "bih"), ("bhh", "bhh")):
        setattr(a, name, dev_ptr(net[key], F32, f"net.{key}").value)
    a.embedded = dev_ptr(emb, F32, "net.embedded").value
    lw = net.get("latent_win")
    a.latent_win = None if lw is None else dev_ptr(lw, F32, "net.latent_win").value
    a.latent_from = -1
    a.sample = int(bool(net.get("sample", False)))                 # draw every pick from the window softmax (stream of sample_seed)
    a.sample_seed = int(net.get("sample_seed", 0)) & 0xFFFFFFFFFFFFFFFF
    a.idx = dev_ptr(out["idx"], I32, "idx").value
    a.win_logits = dev_ptr(out["win_logits"], F32, "win").value
    a.pick_prob = dev_ptr(out["pick_prob"], F32, "prob").value
    a.actions = dev_ptr(out["actions"], F32, "actions").value
    a.queries = None if out["queries"] is None else dev_ptr(out["queries"], F32, "queries").value
    at = _lib.Attention()
    at.attention, at.n_glimpses = ATTENTION_NAMES[attention], int(n_glimpses)
    if attention == "Bahdanau":
        sides = [("pointer", pointer)] + ([("glimpse", glimpse)] if n_glimpses > 0 else [])
        for side, d in sides:
            if d is None:
                raise GnnpnError(f"pointer_decode_attn: 'Bahdanau' needs the {side} module's parameters")
            wq = d["wq"].detach().float().contiguous()
            ref = linear(net["enc_out"].reshape(B * L, H), d["wref"].detach().float().reshape(H, H).contiguous(),
                         d["bref"].detach().float().contiguous()).view(B, L, H)
            bq, v = d["bq"].detach().float().contiguous(), d["v"].detach().float().contiguous()
            keep += [wq, ref, bq, v]
            setattr(at, f"{side}_wq", dev_ptr(wq, F32, f"{side}.wq").value)
            setattr(at, f"{side}_bq", dev_ptr(bq, F32, f"{side}.bq").value)
            setattr(at, f"{side}_ref", dev_ptr(ref, F32, f"{side}.ref").value)
            setattr(at, f"{side}_v", dev_ptr(v, F32, f"{side}.v").value)
    check(_lib.load().gnnpn_pointer_decode_attn_f32(_lib.ctypes.byref(a), _lib.ctypes.byref(at), dev_ptr(inputs, F32, "inputs"),
                                                     float(tanh_c), int(bool(use_tanh)), B, n_cat, n_per, H, stream_ptr()),
          "gnnpn_pointer_decode_attn_f32")
    out["_keep"] = keep + [emb]
    return out


def attention_logits(enc_out, queries, step, idx, tanh_c=10.0, use_tanh=True):
    """Full [B,L] logits of decode step ``step`` with -inf at the ``step`` previously chosen
    positions (API-compat path, see gnnpn_attention_logits_f32)."""
    B, L, H = enc_out.shape
    T = queries.shape[1]
    out = torch.empty((B, L), dtype=F32, device=enc_out.device)
    q = queries[:, step, :]
    check(_lib.load().gnnpn_attention_logits_f32(
        dev_ptr(enc_out, F32, "enc_out"), _lib.ctypes.c_void_p(q.data_ptr()), T * H, dev_ptr(idx, I32, "idx"),
        float(tanh_c), int(bool(use_tanh)), dev_ptr(out, F32, "logits"), B, L, H, step, T, stream_ptr()),
        "gnnpn_attention_logits_f32")
    return out


def attention_logits_bahdanau(ref, qp, v, step, idx, tanh_c=10.0, use_tanh=True):
    """The 'Bahdanau' form of attention_logits (modelPN.py:103-109): ref [B,L,H] = W_ref(enc_out) + b_ref, qp [B,H] = W_query q +
    b_query of the step's pointer query, v [H]; -inf at the ``step`` previously chosen positions."""
    B, L, H = ref.shape
    out = torch.empty((B, L), dtype=F32, device=ref.device)
    check(_lib.load().gnnpn_attention_logits_bahdanau_f32(
        dev_ptr(ref, F32, "ref"), dev_ptr(qp, F32, "qp"), H, dev_ptr(v, F32, "v"), dev_ptr(idx, I32, "idx"), float(tanh_c),
        int(bool(use_tanh)), dev_ptr(out, F32, "logits"), B, L, H, step, idx.shape[1], stream_ptr()),
        "gnnpn_attention_logits_bahdanau_f32")
    return out


def qos_reward(actions, level):
    """actions [B,T,8] -> R [B]; level 'Low' -> #violations, 'High' -> round(violations + objective, 5)."""
    B, T, _ = actions.shape
    R = torch.empty(B, dtype=F32, device=actions.device)
    check(_lib.load().gnnpn_qos_reward_f32(dev_ptr(actions, F32, "actions"), dev_ptr(R, F32, "R"), B, T,
                                           0 if level == "Low" else 1, stream_ptr()), "gnnpn_qos_reward_f32")
    return R


def debug_cell_activations(x):
    """(sigmoid, tanh) as the LSTM-cell kernels evaluate them (test hook)."""
    sig, th = torch.empty_like(x), torch.empty_like(x)
    check(_lib.load().gnnpn_debug_cell_activations(dev_ptr(x, F32, "x"), dev_ptr(sig, F32, "sig"),
                                                   dev_ptr(th, F32, "th"), x.numel(), stream_ptr()),
          "gnnpn_debug_cell_activations")
    return sig, th


def eswoa(cand_ptr, len_init, cand, bounds, start_pos, pop, max_iter, seeds, n_cat, wide=None):
    """ES-WOA fine-tuning of P problems in one launch (gnnpn_eswoa_f64 for T <= 64 categories, gnnpn_eswoa_wide_f64 above;
    reference src/baselines/WOA.py:8-162).
    cand_ptr [P*T+1] i32, len_init [P*T] i32, cand [n,4] f64, bounds [P,4] f64, start_pos [P*T] i32 (first entry of a
    problem < 0: no seed solution), seeds [P] int64 (bit pattern of the uint64 seed); ``wide`` True forces the any-T kernel
    for T <= 64 as well (tests: both forms give the same run).  Returns (best_fitness [P] f64,
    best_pos [P,T] i32, history [P,max_iter] f64, draws [P] i64)."""
    dev = cand.device
    T = int(n_cat)
    P = (cand_ptr.numel() - 1) // T
    I64 = torch.int64
    best_fit = torch.empty(P, dtype=torch.float64, device=dev)
    best_pos = torch.empty(P, T, dtype=I32, device=dev)
    history = torch.empty(P, max(int(max_iter), 1), dtype=torch.float64, device=dev)
    draws = torch.empty(P, dtype=I64, device=dev)
    if T > 64 or wide:      # one workgroup per problem, positions in a workspace (csrc/woa.hip: eswoa_wide_kernel)
        lib = _lib.load()
        nbytes = int(lib.gnnpn_eswoa_wide_workspace_bytes(P, T, int(pop)))
        ws = torch.empty(max(nbytes // 4, 1), dtype=I32, device=dev)
        check(lib.gnnpn_eswoa_wide_f64(P, T, dev_ptr(cand_ptr, I32, "cand_ptr"), dev_ptr(len_init, I32, "len_init"),
                                       dev_ptr(cand, torch.float64, "cand"), dev_ptr(bounds, torch.float64, "bounds"),
                                       dev_ptr(start_pos, I32, "start_pos"), int(pop), int(max_iter),
                                       dev_ptr(seeds, I64, "seeds"), dev_ptr(ws, I32, "workspace"), nbytes,
                                       dev_ptr(best_fit, torch.float64, "best_fitness"), dev_ptr(best_pos, I32, "best_pos"),
                                       dev_ptr(history, torch.float64, "history"), dev_ptr(draws, I64, "draws"), stream_ptr()),
              "gnnpn_eswoa_wide_f64")
        return best_fit, best_pos, history[:, :int(max_iter)], draws
    per_problem = cand_ptr[T::T] - cand_ptr[:-1:T] if P else cand_ptr[:0]
    max_cand = int(per_problem.max().item()) if P else 1
    check(_lib.load().gnnpn_eswoa_f64(P, T, dev_ptr(cand_ptr, I32, "cand_ptr"), dev_ptr(len_init, I32, "len_init"),
                                      dev_ptr(cand, torch.float64, "cand"), dev_ptr(bounds, torch.float64, "bounds"),
                                      dev_ptr(start_pos, I32, "start_pos"), int(pop), int(max_iter),
                                      dev_ptr(seeds, I64, "seeds"), max_cand, dev_ptr(best_fit, torch.float64, "best_fitness"),
                                      dev_ptr(best_pos, I32, "best_pos"), dev_ptr(history, torch.float64, "history"),
                                      dev_ptr(draws, I64, "draws"), stream_ptr()), "gnnpn_eswoa_f64")
    return best_fit, best_pos, history[:, :int(max_iter)], draws


# ---- REINFORCE training step of the High-level pointer network (csrc/train.hip; include/gnnpn_hip.h) -----------------

def gemm(a, b, a_kmajor=False, b_kmajor=False):
    """C[m,n] = sum_k Aop[m,k] * Bop[n,k]; an operand given k-major is [K, M] (resp. [K, N])   (gnnpn_gemm_f32)."""
    a, b = _rows2d(a, "gemm.a"), _rows2d(b, "gemm.b")
    M, K = (a.shape[1], a.shape[0]) if a_kmajor else a.shape
    N, Kb = (b.shape[1], b.shape[0]) if b_kmajor else b.shape
    if K != Kb:
        raise GnnpnError(f"gemm: K mismatch {tuple(a.shape)} x {tuple(b.shape)}")
    tiles = -(-M // 64) * -(-N // 64)
    split = 1 if tiles >= 256 or K < 2048 else max(1, min(64, 512 // tiles, K // 512))   # fill the chip when K >> M, N
    c = torch.empty((split, M, N) if split > 1 else (M, N), dtype=F32, device=a.device)
    check(_lib.load().gnnpn_gemm_f32(dev_ptr(a, F32, "a"), a.shape[1], int(a_kmajor), dev_ptr(b, F32, "b"), b.shape[1],
                                     int(b_kmajor), dev_ptr(c, F32, "c"), N, M, N, K, split, stream_ptr()), "gnnpn_gemm_f32")
    if split > 1:                                         # the partial matrices summed in slice order
        c = colsum(c.view(split, M * N)).view(M, N)
    return c


def colsum(x, rows=None, cols=None, ld=None):
    """out[c] = sum_r x[r, c] (bias gradients; with rows/cols/ld a strided view of a larger buffer)   (gnnpn_colsum_f32)."""
    rows = x.shape[0] if rows is None else rows
    cols = x.shape[-1] if cols is None else cols
    ld = x.shape[-1] if ld is None else ld
    if rows >= 4096:                                      # two passes: the first fills the chip
        per = -(-rows // min(128, rows // 256))
        chunks = -(-rows // per)
        partial = torch.empty((chunks, cols), dtype=F32, device=x.device)
        check(_lib.load().gnnpn_colsum_chunks_f32(dev_ptr(x, F32, "x"), ld, rows, cols, per, dev_ptr(partial, F32, "partial"),
                                                  stream_ptr()), "gnnpn_colsum_chunks_f32")
        x, rows, ld = partial, chunks, cols
    out = torch.empty(cols, dtype=F32, device=x.device)
    check(_lib.load().gnnpn_colsum_f32(dev_ptr(x, F32, "x"), ld, rows, cols, dev_ptr(out, F32, "out"), stream_ptr()),
          "gnnpn_colsum_f32")
    return out


def lstm_train_forward(pregates, whh, bhh):
    """Encoder recurrence that also saves the pre-activation gates and cell states: -> (enc_out, gates_pre, c_all)."""
    B, L, H4 = pregates.shape
    H = H4 // 4
    dev = pregates.device
    enc, gp, ca = (torch.empty(s, dtype=F32, device=dev) for s in ((B, L, H), (B, L, 4 * H), (B, L, H)))
    check(_lib.load().gnnpn_lstm_train_forward_f32(dev_ptr(pregates, F32, "pregates"), dev_ptr(whh, F32, "whh"),
                                                   dev_ptr(bhh, F32, "bhh"), dev_ptr(enc, F32, "enc"), dev_ptr(gp, F32, "gp"),
                                                   dev_ptr(ca, F32, "c"), B, L, H, stream_ptr()), "gnnpn_lstm_train_forward_f32")
    return enc, gp, ca


def _decode_train_struct(d):
    t = _lib.DecodeTrain()
    for name, _ in _lib.DecodeTrain._fields_:
        v = d.get(name)
        setattr(t, name, None if v is None else v.data_ptr())
    return t


def decode_train_forward(embedded, enc_out, h0, c0, start, wih, whh, bih, bhh, latent_win, idx, n_cat, n_per, tanh_c=10.0,
                         use_tanh=True):
    """Teacher-forced decode (picks ``idx`` [B,T] int32 given) that saves what the backward needs; returns the dict of
    operands + saves that decode_train_backward takes, with ``logp`` [B,T] = log-probability of every pick."""
    B, L, H = enc_out.shape
    dev = enc_out.device
    d = {"embedded": embedded, "enc_out": enc_out, "h0": h0, "c0": c0, "start": start, "wih": wih, "whh": whh, "bih": bih,
         "bhh": bhh, "latent_win": latent_win, "idx": idx}
    for name, shape in (("x_all", (B, n_cat, H)), ("gates_pre", (B, n_cat, 4 * H)), ("c_all", (B, n_cat, H)),
                        ("h_all", (B, n_cat, H)), ("z0", (B, n_cat, n_per)), ("probs", (B, n_cat, n_per)), ("logp", (B, n_cat))):
        d[name] = torch.empty(shape, dtype=F32, device=dev)
    for k, v in d.items():
        if v is not None:
            dev_ptr(v, I32 if k == "idx" else F32, k)        # validation (device, dtype, contiguity)
    check(_lib.load().gnnpn_decode_train_forward_f32(_lib.ctypes.byref(_decode_train_struct(d)), B, n_cat, n_per, H,
                                                     float(tanh_c), int(bool(use_tanh)), stream_ptr()),
          "gnnpn_decode_train_forward_f32")
    d.update(n_cat=n_cat, n_per=n_per, tanh_c=float(tanh_c), use_tanh=bool(use_tanh))
    return d


def decode_train_backward(d, gscale):
    """-> (d_enc_out [B,L,H], dgates [B,T,4H], dx [B,T,H], dh0, dc0 [B,H])."""
    B, L, H = d["enc_out"].shape
    T, K = d["n_cat"], d["n_per"]
    dev = d["enc_out"].device
    de, dg, dx, dh0, dc0 = (torch.empty(s, dtype=F32, device=dev) for s in ((B, L, H), (B, T, 4 * H), (B, T, H), (B, H), (B, H)))
    tensors = {k: v for k, v in d.items() if isinstance(v, torch.Tensor) or v is None}
    check(_lib.load().gnnpn_decode_train_backward_f32(
        _lib.ctypes.byref(_decode_train_struct(tensors)), dev_ptr(gscale, F32, "gscale"), dev_ptr(de, F32, "d_enc_out"),
        dev_ptr(dg, F32, "dgates"), dev_ptr(dx, F32, "dx"), dev_ptr(dh0, F32, "dh0"), dev_ptr(dc0, F32, "dc0"), B, T, K, H,
        d["tanh_c"], int(d["use_tanh"]), stream_ptr()), "gnnpn_decode_train_backward_f32")
    return de, dg, dx, dh0, dc0


def _decode_attn_struct(d):
    t = _lib.DecodeAttnTrain()
    t.base = _decode_train_struct(d)
    t.bahdanau, t.n_glimpses = int(d["bahdanau"]), int(d["n_glimpses"])
    for name, _ in _lib.DecodeAttnTrain._fields_[3:]:
        v = d.get(name)
        setattr(t, name, None if v is None else v.data_ptr())
    return t


def decode_attn_train_forward(embedded, enc_out, h0, c0, start, wih, whh, bih, bhh, latent_win, idx, n_cat, n_per, attention,
                              n_glimpses, pointer, glimpse, tanh_c=10.0, use_tanh=True):
    """decode_train_forward through 'Bahdanau' attention and / or glimpse rounds (modelPN.py:80-90,103-109,208-211).
    pointer / glimpse: dicts {wq [H,H], bq [H], wref [H,H,1], bref [H], v [H]} of the two Attention modules ('Bahdanau'; ignored
    for 'Dot').  Returns the dict decode_attn_train_backward takes."""
    B, L, H = enc_out.shape
    dev = enc_out.device
    bah = attention == "Bahdanau"
    if attention not in ("Dot", "Bahdanau"):
        raise NotImplementedError(f"attention '{attention}' (modelPN.py:116-117)")
    d = {"embedded": embedded, "enc_out": enc_out, "h0": h0, "c0": c0, "start": start, "wih": wih, "whh": whh, "bih": bih,
         "bhh": bhh, "latent_win": latent_win, "idx": idx, "bahdanau": bah, "n_glimpses": int(n_glimpses)}
    G = int(n_glimpses)
    for name, shape in (("x_all", (B, n_cat, H)), ("gates_pre", (B, n_cat, 4 * H)), ("c_all", (B, n_cat, H)),
                        ("h_all", (B, n_cat, H)), ("z0", (B, n_cat, n_per)), ("probs", (B, n_cat, n_per)), ("logp", (B, n_cat)),
                        ("q_all", (B, n_cat, G + 1, H))):
        d[name] = torch.empty(shape, dtype=F32, device=dev)
    if G:
        d["a_all"] = torch.empty((B, n_cat, G, L), dtype=F32, device=dev)
    if bah:
        flat = enc_out.reshape(B * L, H)
        for tag, side in (("p", pointer),) + ((("g", glimpse),) if G else ()):
            wq = side["wq"].detach().float().contiguous()
            d[tag + "_wq"], d[tag + "_wq_t"] = wq, wq.t().contiguous()
            d[tag + "_bq"], d[tag + "_v"] = side["bq"].detach().float().contiguous(), side["v"].detach().float().contiguous()
            d[tag + "_wref"] = side["wref"].detach().float().reshape(H, H).contiguous()
            d[tag + "_ref"] = linear(flat, d[tag + "_wref"], side["bref"].detach().float().contiguous()).view(B, L, H)   # :106
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            dev_ptr(v, I32 if k == "idx" else F32, k)        # validation (device, dtype, contiguity)
    check(_lib.load().gnnpn_decode_attn_train_forward_f32(_lib.ctypes.byref(_decode_attn_struct(d)), B, n_cat, n_per, H,
                                                          float(tanh_c), int(bool(use_tanh)), stream_ptr()),
          "gnnpn_decode_attn_train_forward_f32")
    d.update(n_cat=n_cat, n_per=n_per, tanh_c=float(tanh_c), use_tanh=bool(use_tanh))
    return d


def decode_attn_train_backward(d, gscale):
    """-> (d_enc_out [B,L,H] — the part that does not pass through `ref` —, dgates [B,T,4H], dx [B,T,H], dh0, dc0 [B,H]); with
    'Bahdanau' attention d additionally holds d_p_ref / d_g_ref [B,L,H], d_p_qp [B,T,H] / d_g_qp [B,T,G,H], d_p_v / d_g_v [B,H]."""
    B, L, H = d["enc_out"].shape
    T, K, G = d["n_cat"], d["n_per"], d["n_glimpses"]
    dev = d["enc_out"].device
    de = torch.zeros((B, L, H), dtype=F32, device=dev)                   # added to, step by step
    dg, dx, dh0, dc0 = (torch.empty(s, dtype=F32, device=dev) for s in ((B, T, 4 * H), (B, T, H), (B, H), (B, H)))
    if d["bahdanau"]:
        d["d_p_ref"], d["d_p_qp"], d["d_p_v"] = (torch.zeros(s, dtype=F32, device=dev) for s in ((B, L, H), (B, T, H), (B, H)))
        if G:
            d["d_g_ref"], d["d_g_qp"], d["d_g_v"] = (torch.zeros(s, dtype=F32, device=dev) for s in ((B, L, H), (B, T, G, H), (B, H)))
    check(_lib.load().gnnpn_decode_attn_train_backward_f32(
        _lib.ctypes.byref(_decode_attn_struct(d)), dev_ptr(gscale, F32, "gscale"), dev_ptr(de, F32, "d_enc_out"),
        dev_ptr(dg, F32, "dgates"), dev_ptr(dx, F32, "dx"), dev_ptr(dh0, F32, "dh0"), dev_ptr(dc0, F32, "dc0"), B, T, K, H,
        d["tanh_c"], int(d["use_tanh"]), stream_ptr()), "gnnpn_decode_attn_train_backward_f32")
    return de, dg, dx, dh0, dc0


def lstm_train_backward(whh, gates_pre, c_all, d_enc_out, dh0, dc0):
    B, L, H = c_all.shape
    dg = torch.empty((B, L, 4 * H), dtype=F32, device=c_all.device)
    check(_lib.load().gnnpn_lstm_train_backward_f32(dev_ptr(whh, F32, "whh"), dev_ptr(gates_pre, F32, "gates_pre"),
                                                    dev_ptr(c_all, F32, "c_all"), dev_ptr(d_enc_out, F32, "d_enc_out"),
                                                    dev_ptr(dh0, F32, "dh0"), dev_ptr(dc0, F32, "dc0"), dev_ptr(dg, F32, "dgates"),
                                                    B, L, H, stream_ptr()), "gnnpn_lstm_train_backward_f32")
    return dg


def scatter_dx(dx, idx, d_embedded):
    B, T, H = dx.shape
    L = d_embedded.shape[1]
    check(_lib.load().gnnpn_scatter_dx_f32(dev_ptr(dx, F32, "dx"), dev_ptr(idx, I32, "idx"), dev_ptr(d_embedded, F32, "d_embedded"),
                                           B, T, L, H, stream_ptr()), "gnnpn_scatter_dx_f32")


def grad_sumsq(grads):
    """Squared L2 norm over a list of gradient tensors, accumulated on the device in float64 -> tensor [1] f64."""
    acc = torch.zeros(1, dtype=F64, device=grads[0].device)
    for g in grads:
        check(_lib.load().gnnpn_sumsq_f32(dev_ptr(g, F32, "grad"), g.numel(), dev_ptr(acc, F64, "acc"), stream_ptr()),
              "gnnpn_sumsq_f32")
    return acc


# ---- training step of the GNN model (trainML.py; csrc/train_ml.hip) ----------------------------------------------------
def bn_train_forward(x, gamma, beta, relu, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    """BatchNorm1d on batch statistics [+ ReLU]: -> (y, xhat, invstd); the running buffers are updated in place."""
    x = _rows2d(x, "bn_train_forward.x")
    y, xhat = torch.empty_like(x), torch.empty_like(x)
    invstd = torch.empty(x.shape[1], dtype=F32, device=x.device)
    check(_lib.load().gnnpn_bn_train_forward_f32(
        dev_ptr(x, F32, "x"), x.shape[0], x.shape[1], dev_ptr(gamma, F32, "gamma"), dev_ptr(beta, F32, "beta"), float(eps),
        float(momentum), int(bool(relu)), dev_ptr(y, F32, "y"), dev_ptr(xhat, F32, "xhat"), dev_ptr(invstd, F32, "invstd"),
        dev_ptr(running_mean, F32, "running_mean", True), dev_ptr(running_var, F32, "running_var", True), stream_ptr()),
        "gnnpn_bn_train_forward_f32")
    return y, xhat, invstd


def bn_train_backward(dy, y, xhat, gamma, invstd, relu):
    """-> (dx, dgamma, dbeta) of bn_train_forward; dy is the gradient wrt its (post-ReLU) output."""
    dy = _rows2d(dy, "bn_train_backward.dy")
    dx = torch.empty_like(dy)
    dg, db = torch.empty(dy.shape[1], dtype=F32, device=dy.device), torch.empty(dy.shape[1], dtype=F32, device=dy.device)
    check(_lib.load().gnnpn_bn_train_backward_f32(
        dev_ptr(dy, F32, "dy"), dev_ptr(y, F32, "y", True), dev_ptr(xhat, F32, "xhat"), dev_ptr(gamma, F32, "gamma"),
        dev_ptr(invstd, F32, "invstd"), dy.shape[0], dy.shape[1], int(bool(relu)), dev_ptr(dx, F32, "dx"), dev_ptr(dg, F32, "dg"),
        dev_ptr(db, F32, "db"), stream_ptr()), "gnnpn_bn_train_backward_f32")
    return dx, dg, db


def bce_sigmoid(p, y):
    """BCELoss(mean)(p, y) for p = sigmoid(z) -> (loss [1], dLoss/dz, same shape as p)."""
    p = p.contiguous()
    y = y.reshape(p.shape).contiguous()
    dz, loss = torch.empty_like(p), torch.empty(1, dtype=F32, device=p.device)
    check(_lib.load().gnnpn_bce_sigmoid_f32(dev_ptr(p, F32, "p"), dev_ptr(y, F32, "y"), p.numel(), dev_ptr(dz, F32, "dz"),
                                            dev_ptr(loss, F32, "loss"), stream_ptr()), "gnnpn_bce_sigmoid_f32")
    return loss, dz


def dot(a, b):
    a, b = a.contiguous(), b.contiguous()
    if a.numel() != b.numel():
        raise GnnpnError("dot: sizes differ")
    out = torch.empty(1, dtype=F32, device=a.device)
    check(_lib.load().gnnpn_dot_f32(dev_ptr(a, F32, "a"), dev_ptr(b, F32, "b"), a.numel(), dev_ptr(out, F32, "out"), stream_ptr()),
          "gnnpn_dot_f32")
    return out


def embed_grad(dh, x, c, vocab):
    """Gradient of the embedding table of embed_concat: dh [N, >= c] (its first c columns), ids in x[:, 0] -> [vocab, c]."""
    dh, x = _rows2d(dh, "embed_grad.dh"), _rows2d(x, "embed_grad.x")
    out = torch.empty((vocab, c), dtype=F32, device=dh.device)
    check(_lib.load().gnnpn_embed_grad_f32(dev_ptr(dh, F32, "dh"), dh.shape[1], dev_ptr(x, F32, "x"), x.shape[1], dh.shape[0], c,
                                           vocab, dev_ptr(out, F32, "dtable"), stream_ptr()), "gnnpn_embed_grad_f32")
    return out


def adam_step(p, g, m, v, sumsq, max_grad_norm, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """clip_grad_norm_ + torch.optim.Adam (defaults) on one parameter tensor, in place   (gnnpn_adam_step_f32)."""
    check(_lib.load().gnnpn_adam_step_f32(dev_ptr(p, F32, "p"), dev_ptr(g, F32, "g"), dev_ptr(m, F32, "m"), dev_ptr(v, F32, "v"),
                                          p.numel(), dev_ptr(sumsq, F64, "sumsq"), float(max_grad_norm), float(lr), float(beta1),
                                          float(beta2), float(eps), int(step), stream_ptr()), "gnnpn_adam_step_f32")


def split3_pieces(x, scale_log2=0):
    """The three fp16 pieces (as int16 bit patterns, [3, n]) of x * 2^scale_log2 as the recurrent kernels form them
    (GNNPN_PREC_SPLIT; csrc/coop_common.h::split3)."""
    x = x.contiguous().view(-1)
    out = torch.empty((3, x.numel()), dtype=torch.int16, device=x.device)
    check(_lib.load().gnnpn_split3_pieces_f32(dev_ptr(x, F32, "x"), x.numel(), int(scale_log2), out[0].data_ptr(), out[1].data_ptr(),
                                              out[2].data_ptr(), stream_ptr()), "gnnpn_split3_pieces_f32")
    return out


def recurrent_product(whh_packed, h, precision="f32"):
    """gates [16, 4H] = h [16, H] . W_hh^T (H = 256; W_hh packed by pack_lstm_weight) by the fp32 MFMA chain or the exact
    split of the cooperative kernels -> (gates, col_inv [4H] or None)."""
    if h.shape != (16, 256) or whh_packed.numel() != 4 * 256 * 256:
        raise GnnpnError("recurrent_product: h [16,256] and a packed [64,4,256,4] W_hh expected")
    h, whh_packed = h.contiguous(), whh_packed.contiguous()
    gates = torch.empty((16, 1024), dtype=F32, device=h.device)
    prec = _PRECISIONS[precision]
    inv = torch.empty(1024, dtype=F32, device=h.device) if prec == 2 else None
    check(_lib.load().gnnpn_recurrent_product_f32(dev_ptr(whh_packed, F32, "whh"), dev_ptr(h, F32, "h"), prec, dev_ptr(gates, F32, "gates"),
                                                  inv.data_ptr() if inv is not None else None, stream_ptr()), "gnnpn_recurrent_product_f32")
    return gates, inv
