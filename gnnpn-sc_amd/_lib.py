"""ctypes binding of the C ABI declared in ``include/gnnpn_hip.h`` (libgnnpn_hip.so).

There is NO fallback: if the library is missing, or an operand is not a contiguous CUDA tensor of
the right dtype, the call raises.  PyTorch is used only for device memory and the current HIP
stream; tensors cross the boundary as raw device pointers + sizes.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GNNPN_LIB: another build of the same library (the timing-only ablation builds of tools/ablate_aggregate.py); never set in a measured run
LIB_PATH = os.environ.get("GNNPN_LIB") or os.path.join(_HERE, "libgnnpn_hip.so")
ABI_VERSION = 9

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2

_P = c_void_p
_SIGNATURES = {
    "gnnpn_abi_version": (c_int, []),
    "gnnpn_last_error": (c_char_p, []),
    "gnnpn_linear_f32": (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P, c_int, _P, c_int64, c_int64, c_int, c_int, _P]),
    "gnnpn_embed_concat_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_int64, _P]),
    "gnnpn_csr_aggregate_f32": (c_int, [_P, _P, _P, _P, c_int64, _P, _P, _P, _P, c_int, _P, c_int64, c_int32,
                                        c_int32, _P]),
    "gnnpn_csr_aggregate_blocks_f32": (c_int, [_P, _P, _P, _P, c_int64, _P, _P, _P, _P, c_int, _P, c_int64, c_int32,
                                               c_int32, c_int32, _P, _P]),
    "gnnpn_csr_block_row_order": (c_int, [_P, c_int32, c_int32, _P, _P]),
    "gnnpn_csr_tile_plan_geometry": (c_int, [c_int32, c_int32, _P]),
    "gnnpn_csr_tile_plan_rows": (c_int, [_P, _P, _P, c_int32, c_int32, _P, _P, _P, _P, _P, _P]),
    "gnnpn_csr_tile_plan_fill": (c_int, [_P, _P, c_int32, c_int32, _P, _P, _P, _P, c_int64, _P]),
    "gnnpn_csr_aggregate_tiled_f32": (c_int, [_P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, c_int, _P, c_int64, c_int32, c_int32,
                                              c_int32, _P]),
    "gnnpn_gcn_norm_f32": (c_int, [_P, _P, _P, _P, _P, c_int32, _P]),
    "gnnpn_segment_mean_f32": (c_int, [_P, _P, c_int64, _P, c_int64, c_int32, c_int32, _P]),
    "gnnpn_gin_layer_f32": (c_int, [_P, _P, _P, c_int64, c_int32, _P, _P, _P, _P, _P, c_int32, _P, _P, _P, _P, c_int32, _P, _P, c_int32,
                                    _P, c_int64, c_int64, _P]),
    "gnnpn_split_weights_bytes": (c_int64, [c_int32, c_int32]),
    "gnnpn_lstm_split_weights_bytes": (c_int64, []),
    "gnnpn_lstm_pack_split_weights_f32": (c_int, [_P, _P, _P]),
    "gnnpn_pack_split_weights_f16": (c_int, [_P, c_int64, c_int32, c_int32, _P, _P, _P]),
    "gnnpn_gin_layer_split": (c_int, [_P, _P, _P, c_int64, c_int32, _P, _P, _P, _P, _P, _P, c_int32, _P, _P, _P, _P, _P, c_int32,
                                      _P, _P, _P, c_int32, _P, c_int64, c_int64, _P]),
    "gnnpn_request_branch_f32": (c_int, [_P, c_int32, _P, c_int32, c_int32, _P, _P, _P, c_int32, c_int32, c_int32, _P, c_int32,
                                         _P, _P, _P, _P]),
    "gnnpn_select_candidates": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P]),
    "gnnpn_rank_rows": (c_int, [_P, c_int64, _P, c_int32, c_int32, _P]),
    "gnnpn_precision_at_k": (c_int, [_P, c_int64, _P, c_int64, c_int32, c_int32, _P, c_int32, _P, _P]),
    "gnnpn_lstm_encode_f32": (c_int, [c_int, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P, c_int64, _P]),
    "gnnpn_lstm_encode_workspace_bytes": (c_int64, []),
    "gnnpn_set_option": (c_int, [c_char_p, c_int]),
    "gnnpn_decode_diag": (c_int, [_P, c_int32, c_int32]),
    "gnnpn_coop_reset_staffing": (c_int, []),
    "gnnpn_last_launch_units": (c_int64, []),
    "gnnpn_lds_footprint_kb": (c_int, [c_int]),
    "gnnpn_coop_staffing_count": (c_int, []),
    "gnnpn_bn_train_forward_f32": (c_int, [_P, c_int64, c_int32, _P, _P, c_float, c_float, c_int, _P, _P, _P, _P, _P, _P]),
    "gnnpn_bn_train_backward_f32": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int32, c_int, _P, _P, _P, _P]),
    "gnnpn_bce_sigmoid_f32": (c_int, [_P, _P, c_int64, _P, _P, _P]),
    "gnnpn_dot_f32": (c_int, [_P, _P, c_int64, _P, _P]),
    "gnnpn_embed_grad_f32": (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P, _P]),
    "gnnpn_pointer_decode_f32": (c_int, [c_int, _P, _P, c_float, c_int, c_int32, c_int32, c_int32, c_int32, c_int32,
                                         _P, _P, c_int64, _P]),
    "gnnpn_pointer_decode_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32]),
    "gnnpn_pointer_decode_attn_f32": (c_int, [_P, _P, _P, c_float, c_int, c_int32, c_int32, c_int32, c_int32, _P]),
    "gnnpn_attention_logits_f32": (c_int, [_P, _P, c_int64, _P, c_float, c_int, _P, c_int32, c_int32, c_int32,
                                           c_int32, c_int32, _P]),
    "gnnpn_attention_logits_bahdanau_f32": (c_int, [_P, _P, c_int64, _P, _P, c_float, c_int, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    "gnnpn_qos_reward_f32": (c_int, [_P, _P, c_int32, c_int32, c_int, _P]),
    "gnnpn_split3_pieces_f32": (c_int, [_P, c_int64, c_int32, _P, _P, _P, _P]),
    "gnnpn_recurrent_product_f32": (c_int, [_P, _P, c_int32, _P, _P, _P]),
    "gnnpn_gemm_f32": (c_int, [_P, c_int64, c_int, _P, c_int64, c_int, _P, c_int64, c_int64, c_int, c_int, c_int, _P]),
    "gnnpn_colsum_chunks_f32": (c_int, [_P, c_int64, c_int64, c_int32, c_int64, _P, _P]),
    "gnnpn_lstm_train_forward_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P]),
    "gnnpn_decode_train_forward_f32": (c_int, [_P, c_int32, c_int32, c_int32, c_int32, c_float, c_int, _P]),
    "gnnpn_decode_train_backward_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_float, c_int, _P]),
    "gnnpn_lstm_train_backward_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P]),
    "gnnpn_decode_attn_train_forward_f32": (c_int, [_P, c_int32, c_int32, c_int32, c_int32, c_float, c_int, _P]),
    "gnnpn_decode_attn_train_backward_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_float, c_int, _P]),
    "gnnpn_colsum_f32": (c_int, [_P, c_int64, c_int64, c_int32, _P, _P]),
    "gnnpn_scatter_dx_f32": (c_int, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    "gnnpn_sumsq_f32": (c_int, [_P, c_int64, _P, _P]),
    "gnnpn_adam_step_f32": (c_int, [_P, _P, _P, _P, c_int64, _P, c_float, c_float, c_float, c_float, c_float, c_int32, _P]),
    "gnnpn_eswoa_f64": (c_int, [c_int32, c_int32, _P, _P, _P, _P, _P, c_int32, c_int32, _P, c_int32, _P, _P, _P, _P, _P]),
    "gnnpn_eswoa_wide_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32]),
    "gnnpn_eswoa_wide_f64": (c_int, [c_int32, c_int32, _P, _P, _P, _P, _P, c_int32, c_int32, _P, _P, c_int64, _P, _P, _P, _P, _P]),
    "gnnpn_debug_cell_activations": (c_int, [_P, _P, _P, c_int64, _P]),
    "gnnpn_debug_lds_interferer": (c_int, [c_int32, c_int32, c_int32, _P]),
    "gnnpn_gate_wait": (c_int, [_P, c_uint32, c_int32, _P]),
    "gnnpn_debug_mfma_f16": (c_int, [_P, _P, _P, _P, _P]),
}
EXPORTS = tuple(_SIGNATURES)

class DecodeNet(ctypes.Structure):
    """gnnpn_decode_net_t of include/gnnpn_hip.h."""
    _fields_ = [(n, _P) for n in ("embedded", "enc_out", "h0", "c0", "start", "wih_packed", "whh_packed", "bih",
                                  "bhh", "latent_win", "emb_w", "emb_b", "xw_fold", "xb_fold", "start_fold", "idx", "win_logits", "pick_prob", "actions",
                                  "queries")] + \
               [("latent_from", c_int32), ("sample", c_int32), ("sample_seed", ctypes.c_uint64), ("whh_split", _P)]


class Attention(ctypes.Structure):
    """gnnpn_attention_t of include/gnnpn_hip.h."""
    _fields_ = [("attention", c_int32), ("n_glimpses", c_int32)] + \
               [(n, _P) for n in ("pointer_wq", "pointer_bq", "pointer_ref", "pointer_v", "glimpse_wq", "glimpse_bq",
                                  "glimpse_ref", "glimpse_v")]


class LaunchOpts(ctypes.Structure):
    """gnnpn_launch_opts_t of include/gnnpn_hip.h (per-call implementation choice / placement / sticky status)."""
    _fields_ = [("impl", c_int32), ("lds_kb", c_int32), ("write_through", c_int32), ("paired_start", c_int32),
                ("sticky_status", _P)]


class TilePlanGeom(ctypes.Structure):
    """gnnpn_tile_plan_geom_t of include/gnnpn_hip.h."""
    _fields_ = [(n, c_int32) for n in ("n_blocks", "src_tiles", "src_tile_rows", "dst_tiles", "dst_tile_rows", "units", "wavefronts",
                                       "passes")] + \
               [(n, c_int64) for n in ("header_bytes", "order_bytes", "tstart_bytes", "selfw_bytes", "meta_bytes")]


class GinLayer(ctypes.Structure):
    """gnnpn_gin_layer_t of include/gnnpn_hip.h."""
    _fields_ = [(n, _P) for n in ("w0_packed", "b0", "bn1_scale", "bn1_shift", "w3_packed", "b3", "bn2_scale", "bn2_shift",
                                  "eps")]


class DecodeTrain(ctypes.Structure):
    """gnnpn_decode_train_t of include/gnnpn_hip.h."""
    _fields_ = [(n, _P) for n in ("embedded", "enc_out", "h0", "c0", "start", "wih", "whh", "bih", "bhh", "latent_win", "idx",
                                  "x_all", "gates_pre", "c_all", "h_all", "z0", "probs", "logp")]


class DecodeAttnTrain(ctypes.Structure):
    """gnnpn_decode_attn_train_t of include/gnnpn_hip.h."""
    _fields_ = ([("base", DecodeTrain), ("bahdanau", c_int32), ("n_glimpses", c_int32)]
                + [(n, _P) for n in ("p_wq_t", "p_wq", "p_bq", "p_v", "p_ref", "g_wq_t", "g_wq", "g_bq", "g_v", "g_ref", "q_all", "a_all",
                                     "d_p_ref", "d_g_ref", "d_p_qp", "d_g_qp", "d_p_v", "d_g_v")])


class EncodeNet(ctypes.Structure):
    """gnnpn_encode_net_t of include/gnnpn_hip.h."""
    _fields_ = [(n, _P) for n in ("pregates", "inputs", "w_in", "b_in", "whh_packed", "bhh", "enc_out", "h_n", "c_n", "whh_split")]


_lib = None


class GnnpnError(RuntimeError):
    pass


SOURCE_GROUPS = {
    # the sources a committed counter summary depends on: the kernels it measured and every header they include
    "recurrent": ("lstm_coop.hip", "decode_lean.hip", "decode_coop.hip", "coop_common.h", "common.h", "recurrent.h", "lstm_shared.h",
                  "decode_shared.h", "lstm.hip", "decode.hip"),
    "aggregate": ("graph_tiled.hip", "graph.hip", "graph_lds.h", "common.h"),
}


def source_hash(group=None):
    """sha256 (16 hex digits) over the device sources this library is built from, names and contents in sorted order: all of
    csrc/*.hip, *.h, torch_ops.cpp and include/gnnpn_hip.h, or (``group``: "recurrent" | "aggregate") the files one family of kernels
    is compiled from.  The committed counter summaries under profiles/ (PMC traffic, SQ matrix-pipe utilisation) carry the hash of
    the sources they were measured on; bench.py quotes them beside freshly measured numbers only while it equals the loaded
    tree's (they used to be keyed by workload, batch and precision alone and went stale silently)."""
    import hashlib
    root = os.path.dirname(_HERE)
    csrc = os.path.join(_HERE, "csrc")
    if group is None:
        files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".cpp")))
    else:
        files = sorted(os.path.join(csrc, f) for f in SOURCE_GROUPS[group])
    files.append(os.path.join(root, "include", "gnnpn_hip.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load():
    """Load libgnnpn_hip.so (once) and declare every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GnnpnError(
            f"{LIB_PATH} is missing: build it with `python gnnpn-sc_amd/build.py` (hipcc, gfx950). "
            "There is no CPU fallback for the ML+2PN hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here = the .so does not export the header's symbol
        fn.restype, fn.argtypes = res, args
    if lib.gnnpn_abi_version() != ABI_VERSION:
        raise GnnpnError(f"ABI mismatch: library {lib.gnnpn_abi_version()} vs binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().gnnpn_last_error().decode("utf-8", "replace")
        raise GnnpnError(f"{what} failed ({rc}): {msg}")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev_ptr(t, dtype, name, allow_none=False):
    """Raw device pointer of a contiguous CUDA tensor (validated), or NULL."""
    if t is None:
        if allow_none:
            return None
        raise GnnpnError(f"{name}: tensor required")
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise GnnpnError(f"{name}: expected a CUDA tensor (the hot path has no CPU implementation), got "
                         f"{type(t).__name__} on {getattr(t, 'device', '?')}")
    if t.dtype != dtype:
        raise GnnpnError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise GnnpnError(f"{name}: tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def ptr_array(tensors, dtype, name):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = dev_ptr(t, dtype, f"{name}[{i}]").value
    return arr
