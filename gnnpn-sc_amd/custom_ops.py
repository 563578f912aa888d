"""PyTorch-ROCm custom operators of the ML+2PN path: one ``gnnpn::`` namespace registered with ``torch.library``
(SURVEY.md section 8b "Custom-op layer"), each operator implemented for the CUDA (= HIP on ROCm) dispatch key ONLY by a
call into the C ABI of libgnnpn_hip.so (``ops.py`` -> ``include/gnnpn_hip.h``).  There is no CPU kernel registered:
calling an operator with host tensors fails in the dispatcher ("Could not run 'gnnpn::...' with arguments from the
'CPU' backend"), and a missing library raises ``GnnpnError`` — never a silent fallback.

    torch.ops.gnnpn.linear / embed_concat / csr_aggregate / gcn_norm / segment_mean
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
import torch
from torch.library import Library

from . import ops

_DEF = Library("gnnpn", "DEF")
_IMPL = Library("gnnpn", "IMPL", "CUDA")

# fixed order of the per-net operands of pointer_decode (None = absent)
DECODE_KEYS = ("enc_out", "h0", "c0", "start", "wih", "whh", "bih", "bhh", "embedded", "emb_w", "emb_b",
               "xw_fold", "xb_fold", "start_fold", "latent_win")
DECODE_OUTS = ("idx", "win_logits", "pick_prob", "actions", "queries")
ENCODE_KEYS = ("pregates", "inputs", "w_in", "b_in", "whh", "bhh")


def _op(schema, fn):
    _DEF.define(schema)
    _IMPL.impl(schema.split("(")[0], fn)


def _ws(ws_id):
    return None if ws_id < 0 else ops.workspaces_by_id(ws_id)


_op("linear(Tensor a, Tensor weight, Tensor? bias=None, Tensor? scale=None, Tensor? shift=None, int act=0) -> Tensor",
    lambda a, weight, bias=None, scale=None, shift=None, act=0: ops.linear(a, weight, bias, scale, shift, act))
_op("embed_concat(Tensor x, Tensor table) -> Tensor", ops.embed_concat)
_op("csr_aggregate(Tensor rowptr, Tensor col, Tensor? w, Tensor x, Tensor? self_coef=None, Tensor? bias=None, "
    "Tensor? scale=None, Tensor? shift=None, int act=0, int block_rows=0) -> Tensor",
    lambda rowptr, col, w, x, self_coef=None, bias=None, scale=None, shift=None, act=0, block_rows=0:
    ops.csr_aggregate(rowptr, col, w, x, self_coef, bias, scale, shift, act, block_rows))
def _request_branch(x, table, rowptr, col, seg_ptr, max_nodes, layer_tensors, lin_w_packed, lin_b, hidden):
    keys = ("w0p", "b0", "a1", "s1", "w3p", "b3", "a2", "s2", "eps")
    layers = [dict(zip(keys, layer_tensors[i:i + len(keys)])) for i in range(0, len(layer_tensors), len(keys))]
    return ops.request_branch(x, table, rowptr, col, seg_ptr, max_nodes, layers, lin_w_packed, lin_b, hidden)


_op("request_branch(Tensor x, Tensor table, Tensor rowptr, Tensor col, Tensor seg_ptr, int max_nodes, "
    "Tensor[] layer_tensors, Tensor lin_w_packed, Tensor lin_b, int hidden) -> Tensor", _request_branch)
_op("gin_layer(Tensor rowptr, Tensor col, Tensor x, Tensor eps, Tensor w1, Tensor? b1, Tensor? a1, Tensor? s1, Tensor w2, Tensor? b2, "
    "Tensor? a2, Tensor? s2, Tensor? w3=None, Tensor? b3=None) -> Tensor", ops.gin_layer)
_op("gin_layer_split(Tensor rowptr, Tensor col, Tensor x, Tensor eps, Tensor w1, Tensor i1, Tensor? b1, Tensor? a1, Tensor? s1, "
    "Tensor w2, Tensor i2, Tensor? b2, Tensor? a2, Tensor? s2, Tensor? w3=None, Tensor? i3=None, Tensor? b3=None) -> Tensor",
    ops.gin_layer_split)
_op("gcn_norm(Tensor rowptr, Tensor col, Tensor w_raw) -> Tensor", ops.gcn_norm)
_op("segment_mean(Tensor segptr, Tensor x) -> Tensor", ops.segment_mean)
_op("segment_topk_feasible(Tensor scores, Tensor cat_ptr, Tensor qos, Tensor local_bounds, Tensor present, "
    "Tensor global_bounds, int n_per) -> (Tensor, Tensor)", ops.select_candidates)
_op("rank_rows(Tensor scores) -> Tensor", ops.rank_rows)
_op("precision_at_k(Tensor ranking, Tensor labels, int[] ks) -> Tensor",
    lambda ranking, labels, ks: ops.precision_at_k(ranking, labels, tuple(ks)))
_op("attention_logits(Tensor enc_out, Tensor queries, int step, Tensor idx, float tanh_c, bool use_tanh) -> Tensor",
    ops.attention_logits)
_op("qos_reward(Tensor actions, int level) -> Tensor",
    lambda actions, level: ops.qos_reward(actions, "Low" if level == 0 else "High"))


def _lstm_encode(net_tensors, n_nets, precision="f32", impl=0, lds_kb=0, write_through=False, ws_id=-1, paired_start=False):   # defaults = the schema's: the dispatcher omits arguments that equal them
    n = len(ENCODE_KEYS)
    nets = [{k: net_tensors[i * n + j] for j, k in enumerate(ENCODE_KEYS)} for i in range(n_nets)]
    enc, h_n, c_n = ops.lstm_encode(nets, precision, impl, lds_kb, write_through, _ws(ws_id), paired_start)
    return list(enc) + list(h_n) + list(c_n)


_op("lstm_encode(Tensor?[] net_tensors, int n_nets, str precision='f32', int impl=0, int lds_kb=0, "
    "bool write_through=False, int ws=-1, bool paired_start=False) -> Tensor[]", _lstm_encode)


def _pointer_decode(net_tensors, latent_from, inputs, n_cat, n_per, tanh_c=10.0, use_tanh=True, want_queries=False,
                    precision="f32", impl=0, lds_kb=0, write_through=False, ws_id=-1, sample_seeds=(), paired_start=False):
    n = len(DECODE_KEYS)
    nets = []
    for i, lf in enumerate(latent_from):
        d = {k: net_tensors[i * n + j] for j, k in enumerate(DECODE_KEYS)}
        d["latent_from"] = lf
        if i < len(sample_seeds) and sample_seeds[i] >= 0:      # >= 0: draw this net's picks from the stream of that seed
            d["sample"], d["sample_seed"] = True, sample_seeds[i]
        nets.append(d)
    outs = ops.pointer_decode(nets, inputs, n_cat, n_per, tanh_c, use_tanh, want_queries, precision, impl, lds_kb,
                              write_through, _ws(ws_id), paired_start)
    flat = []
    for o in outs:
        for k in DECODE_OUTS:
            flat.append(o[k] if o[k] is not None else inputs.new_empty(0))
    return flat


_op("pointer_decode(Tensor?[] net_tensors, int[] latent_from, Tensor inputs, int n_cat, int n_per, float tanh_c=10.0, "
    "bool use_tanh=True, bool want_queries=False, str precision='f32', int impl=0, int lds_kb=0, "
    "bool write_through=False, int ws=-1, int[] sample_seeds=[], bool paired_start=False) -> Tensor[]", _pointer_decode)


# ---- thin callers used by the mirrors: dict-of-tensors in, torch.ops.gnnpn.* underneath ------------------------------

def lstm_encode(nets, precision="f32", impl=0, lds_kb=0, write_through=False, ws=None, paired_start=False):
    flat = [d.get(k) for d in nets for k in ENCODE_KEYS]
    out = torch.ops.gnnpn.lstm_encode(flat, len(nets), precision, impl, lds_kb, bool(write_through),
                                      -1 if ws is None else ws.id, bool(paired_start))
    n = len(nets)
    return out[:n], out[n:2 * n], out[2 * n:]


def pointer_decode(nets, inputs, n_cat, n_per, tanh_c=10.0, use_tanh=True, want_queries=False, precision="f32", impl=0,
                   lds_kb=0, write_through=False, ws=None, paired_start=False):
    flat = [d.get(k) for d in nets for k in DECODE_KEYS]
    lf = [int(d.get("latent_from", -1)) for d in nets]
    seeds = [int(d["sample_seed"]) & 0x7FFFFFFFFFFFFFFF if d.get("sample") else -1 for d in nets]
    out = torch.ops.gnnpn.pointer_decode(flat, lf, inputs, n_cat, n_per, float(tanh_c), bool(use_tanh), bool(want_queries),
                                         precision, impl, lds_kb, bool(write_through), -1 if ws is None else ws.id, seeds,
                                         bool(paired_start))
    m = len(DECODE_OUTS)
    res = []
    for i in range(len(nets)):
        d = dict(zip(DECODE_OUTS, out[i * m:(i + 1) * m]))
        if d["queries"].numel() == 0 and not want_queries:
            d["queries"] = None
        res.append(d)
    return res
