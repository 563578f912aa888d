"""Build recipe for libgnnpn_hip.so (explicit hipcc, gfx950 only, in-tree so that the .so travels
to the GPU box with the repo snapshot).

    python gnnpn-sc_amd/build.py [--force]

Flags of note: ``-ffp-contract=off`` — every fused multiply-add in the kernels is an explicit
``fmaf`` and every place that must round a product before adding says ``__fmul_rn/__fadd_rn``;
the compiler is not allowed to change either.  No fast-math.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgnnpn_hip.so")
SOURCES = ["api.hip", "dense.hip", "graph.hip", "graph_tiled.hip", "gin_layer.hip", "gin_layer_split.hip", "request_branch.hip", "select.hip", "lstm.hip", "lstm_coop.hip", "decode.hip", "decode_coop.hip", "decode_lean.hip", "train.hip", "train_attn.hip", "train_ml.hip", "decode_glimpse.hip", "woa.hip", "split_probe.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "graph_lds.h"), os.path.join(CSRC, "recurrent.h"), os.path.join(CSRC, "lstm_shared.h"), os.path.join(CSRC, "decode_shared.h"), os.path.join(CSRC, "coop_common.h"), os.path.join(CSRC, "train_common.h"),
           os.path.join(ROOT, "include", "gnnpn_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + os.environ.get("GNNPN_EXTRA_HIPCC_FLAGS", "").split()   # (experiments only)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + HEADERS):
        subprocess.run(["hipcc"] + FLAGS + ["-c", path, "-o", obj], check=True)
    return obj


# The same-XCD hand-off of the cooperative kernels publishes with plain global stores — since round 6 WRITTEN as that instruction
# (coop_common.h: granule_store_l2_bits, inline asm; decode_lean.hip: raw buffer stores with explicit cache-policy bits), not asked
# for as workgroup-scope atomics that this toolchain happened to lower that way: the form no longer depends on the compiler, and the
# warning rounds 3-5 printed under another hipcc is gone.  The compiler's identity is still recorded beside the objects.
def _check_compiler():
    try:
        out = subprocess.run(["hipcc", "--version"], capture_output=True, text=True, check=True).stdout
    except (OSError, subprocess.CalledProcessError):
        return
    with open(os.path.join(HERE, "build", "compiler.txt"), "w") as f:
        f.write(out)


def build(force=False):
    """Compile every HIP source for gfx950 and link libgnnpn_hip.so.  Returns the library path."""
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    _check_compiler()
    if force:
        for f in os.listdir(os.path.join(HERE, "build")):
            if os.path.isfile(os.path.join(HERE, "build", f)):
                os.remove(os.path.join(HERE, "build", f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if force or _stale(LIB, objs):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, check=True)
    build_torch_ops(force)
    return LIB


TORCH_LIB = os.path.join(HERE, "libgnnpn_torch.so")


def build_torch_ops(force=False):
    """libgnnpn_torch.so: the C++ registration of the ``gnnpn::`` operators (csrc/torch_ops.cpp — TORCH_LIBRARY schemas and CUDA
    implementations over the C ABI).  Host code only: the host compiler against torch's headers, linked to torch's libraries and to
    libgnnpn_hip.so (found beside it through $ORIGIN)."""
    src = os.path.join(CSRC, "torch_ops.cpp")
    if not (force or _stale(TORCH_LIB, [src, os.path.join(ROOT, "include", "gnnpn_hip.h"), LIB])):
        return TORCH_LIB
    import torch
    from torch.utils import cpp_extension
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
            f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations"] +
           ["-I" + p for p in cpp_extension.include_paths()] + ["-I" + os.path.join(rocm, "include"), "-I" + os.path.join(ROOT, "include"),
            src, "-o", TORCH_LIB, "-L" + tlib, "-L" + HERE, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-lgnnpn_hip",
            "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib])
    subprocess.run(cmd, check=True)
    return TORCH_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
