"""The one-launch GIN layer kernels on their own at the 1000-task shape: layer 0 (26 -> 256 -> 128) and layer 1 (128 -> 256 ->
128 -> nodeLin 128), fp32 matrix core (gnnpn_gin_layer_f32) and fp16 matrix cores through the exact split (gnnpn_gin_layer_split).
    python tools/bench_gin_layer.py [--rows 512512] [--nodes 1001]     (GNNPN_LIB=<ablation build> for timing-only builds)"""
import argparse, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import graph, ops

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=512 * 1001)
ap.add_argument("--nodes", type=int, default=1001)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--forms", default="f32,split")
ap.add_argument("--pad0", action="store_true", help="also time layer 0 on the input zero-padded to 32 channels (16-byte gathers)")
a = ap.parse_args()
dev = torch.device("cuda:0")
n = a.rows
g = torch.Generator().manual_seed(3)
i = torch.arange(n - 1)
keep = (i + 1) % a.nodes != 0                                       # chain graphs (the workflow graphs of the synthetic requests)
ei = torch.stack([torch.stack([i[keep], i[keep] + 1], 1).reshape(-1), torch.stack([i[keep] + 1, i[keep]], 1).reshape(-1)])
csr = graph.csr_by_destination(ei, n).to(dev)
mk = lambda *s: (torch.randn(*s, generator=g) / s[-1] ** 0.5).to(dev)   # noqa: E731
eps = torch.tensor([0.0], device=dev)


def timed(fn):
    best = float("inf")
    for rnd in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            best = min(best, e0.elapsed_time(e1) / a.reps)
    return best


out = {"rows": n, "lib": os.path.basename(os.environ.get("GNNPN_LIB", "libgnnpn_hip.so"))}
for name, c_in, lin in (("layer0", 26, False), ("layer1", 128, True)):
    x = mk(n, c_in)
    w1, b1, w2, b2, w3, b3 = mk(256, c_in), mk(256), mk(128, 256), mk(128), mk(128, 128), mk(128)
    a1, s1, a2, s2 = torch.rand(256, device=dev) + 0.5, mk(256), torch.rand(128, device=dev) + 0.5, mk(128)
    flop = n * 2 * (c_in * 256 + 256 * 128 + (128 * 128 if lin else 0))
    if "f32" in a.forms:
        q = [ops.pack_mfma_b32(w) for w in (w1, w2, w3)]
        ms = timed(lambda: ops.gin_layer(csr.rowptr, csr.col, x, eps, q[0], b1, a1, s1, q[1], b2, a2, s2, q[2] if lin else None, b3 if lin else None))
        out[name + "_f32"] = {"ms": round(ms, 4), "TFLOPs": round(flop / ms / 1e9, 1)}
    if "split" in a.forms:
        p = [ops.pack_split_weights(w) for w in (w1, w2, w3)]
        ms = timed(lambda: ops.gin_layer_split(csr.rowptr, csr.col, x, eps, *p[0], b1, a1, s1, *p[1], b2, a2, s2,
                                               *(p[2] if lin else (None, None)), b3 if lin else None))
        out[name + "_split"] = {"ms": round(ms, 4), "TFLOPs_fp32_equivalent": round(flop / ms / 1e9, 1)}
    if a.pad0 and c_in % 4:
        xp = torch.nn.functional.pad(x, (0, 32 - c_in)).contiguous()
        if "split" in a.forms:
            f = lambda xx: ops.gin_layer_split(csr.rowptr, csr.col, xx, eps, *p[0], b1, a1, s1, *p[1], b2, a2, s2, None, None, None)   # noqa: E731
            out[name + "_split_padded32"] = {"ms": round(timed(lambda: f(xp)), 4), "same_bits": bool(torch.equal(f(xp), f(x)))}
        if "f32" in a.forms:
            f = lambda xx: ops.gin_layer(csr.rowptr, csr.col, xx, eps, q[0], b1, a1, s1, q[1], b2, a2, s2, None, None)   # noqa: E731
            out[name + "_f32_padded32"] = {"ms": round(timed(lambda: f(xp)), 4), "same_bits": bool(torch.equal(f(xp), f(x)))}
print(json.dumps(out))
