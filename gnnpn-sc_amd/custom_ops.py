"""PyTorch-ROCm custom operators of the ML+2PN path: one ``gnnpn::`` namespace registered FROM C++ (``csrc/torch_ops.cpp``:
TORCH_LIBRARY schemas + TORCH_LIBRARY_IMPL for the CUDA (= HIP on ROCm) dispatch key, built by ``build.py`` into the in-tree
``libgnnpn_torch.so``; SURVEY.md section 8b "Custom-op layer").  Each operator is a C++ call into the C ABI of libgnnpn_hip.so
(``include/gnnpn_hip.h``) on the current HIP stream.  There is no CPU kernel registered: calling an operator with host tensors
fails in the dispatcher ("Could not run 'gnnpn::...' with arguments from the 'CPU' backend"), and a missing library raises
``GnnpnError`` — never a silent fallback.  This module loads the library and keeps the few callers that turn the mirrors'
dicts of tensors into the operators' flat argument lists (and resolve ``ops.Workspaces`` to the workspace / status tensors the
recurrent operators take).  ``ops.py`` stays the ctypes binding of the same C ABI (tests, tools, the training step).

    torch.ops.gnnpn.linear / embed_concat / gcn_norm / segment_mean
    torch.ops.gnnpn.csr_aggregate (gather) / csr_aggregate_blocks / csr_aggregate_tiled   (form chosen by ``csr_aggregate`` below)
    torch.ops.gnnpn.request_branch             (the whole GIN branch of small workflow graphs in one launch)
    torch.ops.gnnpn.gin_layer                  (one GIN layer of large workflow graphs in one launch)
    torch.ops.gnnpn.gin_layer_split            (the same on the fp16 matrix cores through the exact split)
    torch.ops.gnnpn.segment_topk_feasible      (candidate reduction: sort + loadDataPN + SCDataset)
    torch.ops.gnnpn.rank_rows / precision_at_k
    torch.ops.gnnpn.lstm_encode                (n nets in one launch)
    torch.ops.gnnpn.pointer_decode             (1 or 2 nets in one launch, the two-level scheme)
    torch.ops.gnnpn.attention_logits / qos_reward

The mirrors of the reference's modules (modelML.Net, modelPN.PointerNet / CombinatorialRL, pipeline) call these
operators; operands are borrowed, outputs are allocated by the operator, work is enqueued on the current HIP stream
(so the operators are capturable into HIP graphs).  Inference only: no autograd formulas are registered.
"""
import os

import torch

from . import _lib, ops

DECODE_KEYS = ("enc_out", "h0", "c0", "start", "wih", "whh", "bih", "bhh", "embedded", "emb_w", "emb_b",
               "xw_fold", "xb_fold", "start_fold", "latent_win", "whh_split")
DECODE_OUTS = ("idx", "win_logits", "pick_prob", "actions", "queries")
ENCODE_KEYS = ("pregates", "inputs", "w_in", "b_in", "whh", "bhh", "whh_split")
REQUEST_LAYER_KEYS = ("w0p", "b0", "a1", "s1", "w3p", "b3", "a2", "s2", "eps")

TORCH_LIB_PATH = os.environ.get("GNNPN_TORCH_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgnnpn_torch.so")


def _load():
    """Load libgnnpn_torch.so (csrc/torch_ops.cpp): its static initialisers register the ``gnnpn::`` schemas and the CUDA
    implementations with the dispatcher.  The kernel library is loaded first (same path the ctypes binding uses; an ABI mismatch or
    a missing file raises GnnpnError) — the operator library links to it and finds it beside itself."""
    _lib.load()
    if not os.path.exists(TORCH_LIB_PATH):
        raise _lib.GnnpnError(f"{TORCH_LIB_PATH} is missing: build it with `python gnnpn-sc_amd/build.py` (the gnnpn:: operators are "
                              "registered from C++; there is no Python or CPU fallback)")
    torch.ops.load_library(TORCH_LIB_PATH)


_load()


# ---- thin callers used by the mirrors: dict-of-tensors in, torch.ops.gnnpn.* (C++) underneath ---------------------------

def _gnnpn_errors(fn):
    """An operator's c10::Error (RuntimeError: bad operand, a shape the kernels are not built for, a failed launch) as the
    package's GnnpnError, which the mirrors' callers catch (it IS a RuntimeError)."""
    import functools

    @functools.wraps(fn)
    def call(*a, **k):
        try:
            return fn(*a, **k)
        except _lib.GnnpnError:
            raise
        except RuntimeError as e:
            raise _lib.GnnpnError(str(e).split("\nException raised from")[0]) from None
    return call


def _coop_ws(device, H, n_per, impl, ws):
    """The workspaces a cooperative launch needs (None for shapes that take the streaming form)."""
    return ops.workspaces(device, ws) if ops.coop_supported(H, n_per, impl) else None


@_gnnpn_errors
def lstm_encode(nets, precision="f32", impl=0, lds_kb=0, write_through=False, ws=None, paired_start=False):
    """n nets in one launch (gnnpn_lstm_encode_f32): see ops.lstm_encode for the operands.  -> (enc_out list, h_n list, c_n list)"""
    flat = [d.get(k) for d in nets for k in ENCODE_KEYS]
    first = nets[0]["pregates"] if nets[0].get("pregates") is not None else nets[0]["inputs"]
    wsp = _coop_ws(first.device, nets[0]["bhh"].numel() // 4, 1, impl, ws)
    out = torch.ops.gnnpn.lstm_encode(flat, len(nets), precision, impl, lds_kb, bool(write_through),
                                      None if wsp is None else wsp.encode(), None if wsp is None else wsp.status, bool(paired_start))
    if wsp is not None:
        wsp.note_launch(0, _lib.load().gnnpn_last_launch_units())     # the host's count of the work the launcher asked for
    n = len(nets)
    return out[:n], out[n:2 * n], out[2 * n:]


@_gnnpn_errors
def pointer_decode(nets, inputs, n_cat, n_per, tanh_c=10.0, use_tanh=True, want_queries=False, precision="f32", impl=0,
                   lds_kb=0, write_through=False, ws=None, paired_start=False):
    """1 or 2 nets in one launch (gnnpn_pointer_decode_f32): see ops.pointer_decode.  -> one dict per net (DECODE_OUTS)"""
    flat = [d.get(k) for d in nets for k in DECODE_KEYS]
    lf = [int(d.get("latent_from", -1)) for d in nets]
    seeds = [int(d["sample_seed"]) & 0x7FFFFFFFFFFFFFFF if d.get("sample") else -1 for d in nets]
    B, L, H = nets[0]["enc_out"].shape
    wsp = _coop_ws(inputs.device, H, n_per, impl, ws)
    out = torch.ops.gnnpn.pointer_decode(flat, lf, inputs, n_cat, n_per, float(tanh_c), bool(use_tanh), bool(want_queries),
                                         precision, impl, lds_kb, bool(write_through),
                                         None if wsp is None else wsp.decode(B, n_cat, n_per), None if wsp is None else wsp.status,
                                         seeds, bool(paired_start))
    if wsp is not None:
        wsp.note_launch(1, _lib.load().gnnpn_last_launch_units())
    m = len(DECODE_OUTS)
    res = []
    for i in range(len(nets)):
        d = dict(zip(DECODE_OUTS, out[i * m:(i + 1) * m]))
        if d["queries"].numel() == 0 and not want_queries:
            d["queries"] = None
        res.append(d)
    return res


@_gnnpn_errors
def request_branch(x, table, rowptr, col, seg_ptr, max_nodes, layers, lin_w_packed, lin_b, hidden):
    """The whole GIN branch of small workflow graphs in one launch; ``layers``: list of dicts (REQUEST_LAYER_KEYS)."""
    flat = [lp[k] for lp in layers for k in REQUEST_LAYER_KEYS]
    return torch.ops.gnnpn.request_branch(x, table, rowptr, col, seg_ptr, int(max_nodes), flat, lin_w_packed, lin_b, int(hidden))


@_gnnpn_errors
def csr_aggregate(rowptr, col, w, x, self_coef=None, bias=None, scale=None, shift=None, act=0, block_rows=0):
    """The CSR aggregate with the form chosen per graph by ops.csr_aggregate_form (tiled where the graph's plan is valid,
    whole-block LDS where a block fits with 16-channel slices, else the gather; bit-identical every way) — the policy and the
    per-graph caches (tile plan, row order) live in ops.py, the launch goes through the C++ operators."""
    form, aux = ops.csr_aggregate_form(rowptr, col, w, x, block_rows)
    if form == "tiled":
        return torch.ops.gnnpn.csr_aggregate_tiled(aux.header, aux.order, aux.selfw, aux.batches, x, self_coef, bias, scale, shift, act,
                                                   aux.n_rows, aux.block_rows)
    if form == "blocks":
        return torch.ops.gnnpn.csr_aggregate_blocks(rowptr, col, w, x, self_coef, bias, scale, shift, act, int(block_rows), aux)
    return torch.ops.gnnpn.csr_aggregate(rowptr, col, w, x, self_coef, bias, scale, shift, act)
