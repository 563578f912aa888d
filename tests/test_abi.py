"""-m "not gpu": the C-ABI shared library builds, loads and exports exactly what include/gnnpn_hip.h
declares; the binding fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared():
    with open(os.path.join(ROOT, "include", "gnnpn_hip.h")) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(gnnpn_[a-z0-9_]+)\s*\(", text)))


def test_build_and_load():
    import __graft_entry__ as entry
    lib_path = entry.build()
    assert os.path.exists(lib_path)
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gnnpn_hip.h but not exported"
    lib.gnnpn_abi_version.restype = ctypes.c_int
    assert lib.gnnpn_abi_version() == 9


def test_binding_covers_header():
    from gnnpn_sc_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()
    _lib.load()


def test_argument_validation_without_gpu():
    """Error paths return before any launch, so they can be exercised on a CPU-only box."""
    from gnnpn_sc_amd import _lib
    lib = _lib.load()
    rc = lib.gnnpn_linear_f32(None, 4, None, 4, None, None, None, 0, None, 4, 2, 2, 4, None)
    assert rc == -1 and b"null" in lib.gnnpn_last_error()
    rc = lib.gnnpn_lstm_encode_f32(9, None, 1, 1, 256, 8, 0, None, None, 0, None)
    assert rc == -1 and b"n_nets" in lib.gnnpn_last_error()
    assert lib.gnnpn_set_option(b"decode_impl", 3) == -1        # implementation choice is a per-call argument now
    assert b"gnnpn_launch_opts_t" in lib.gnnpn_last_error()
    rc = lib.gnnpn_rank_rows(ctypes.c_void_p(16), 40000, ctypes.c_void_p(16), 1, 40000, None)
    assert rc == -2                                     # unsupported size is reported, not truncated


def test_no_cpu_fallback():
    from gnnpn_sc_amd import ops
    with pytest.raises(ops.GnnpnError, match="CUDA tensor"):
        ops.linear(torch.rand(4, 8), torch.rand(3, 8))
    with pytest.raises(ops.GnnpnError):
        ops.qos_reward(torch.rand(2, 3, 8), "High")


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from gnnpn_sc_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.GnnpnError, match="no CPU fallback"):
        _lib.load()


GNNPN_OPERATORS = ("linear", "embed_concat", "csr_aggregate", "csr_aggregate_blocks", "csr_aggregate_tiled", "gcn_norm", "segment_mean",
                   "request_branch", "gin_layer", "gin_layer_split", "segment_topk_feasible", "rank_rows", "precision_at_k",
                   "attention_logits", "qos_reward", "lstm_encode", "pointer_decode")


def test_operators_are_registered_from_cpp():
    """The gnnpn:: operators come from libgnnpn_torch.so (csrc/torch_ops.cpp, built in-tree by build.py): importing custom_ops
    loads it, every operator is there with a C++ kernel for the CUDA key and none for the CPU key (host tensors fail in the
    dispatcher — no fallback), and nothing in custom_ops.py registers an implementation from Python."""
    import __graft_entry__ as entry
    entry.build()
    from gnnpn_sc_amd import custom_ops
    assert os.path.exists(custom_ops.TORCH_LIB_PATH) and os.path.dirname(custom_ops.TORCH_LIB_PATH) == os.path.join(ROOT, "gnnpn-sc_amd")
    with open("/proc/self/maps") as f:
        assert "libgnnpn_torch.so" in f.read()
    for name in GNNPN_OPERATORS:
        op = getattr(torch.ops.gnnpn, name)
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"gnnpn::{name}", "CUDA"), name
        assert not torch._C._dispatch_has_kernel_for_dispatch_key(f"gnnpn::{name}", "CPU"), name
        assert op.default._schema.name == f"gnnpn::{name}"
    with pytest.raises((NotImplementedError, RuntimeError), match="CPU"):
        torch.ops.gnnpn.linear(torch.rand(4, 8), torch.rand(3, 8))
    with open(os.path.join(ROOT, "gnnpn-sc_amd", "custom_ops.py")) as f:
        src = f.read()
    assert "torch.library" not in src and ".impl(" not in src and ".define(" not in src


def test_missing_operator_library_is_loud(monkeypatch, tmp_path):
    from gnnpn_sc_amd import _lib, custom_ops
    monkeypatch.setattr(custom_ops, "TORCH_LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.GnnpnError, match="no Python or CPU fallback"):
        custom_ops._load()


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under gnnpn-sc_amd/ may reference it."""
    pkg = os.path.join(ROOT, "gnnpn-sc_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{fn} imports oracle"
