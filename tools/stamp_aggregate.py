"""Phase stamps of the tiled aggregate (round 5): a diagnostic build of the library (tools/experiments/aggregate_stamps.patch on a
copy of csrc/) records, per wavefront of the first 4096 workgroups, the cycle counter at every phase boundary — entry, headers,
per source tile {own part of the fill written, barrier passed, own units done}, epilogue arithmetic done, stores issued, stores
acknowledged.  Where a workgroup's 49 us at 5000 rows x 128 copies go, and how far the wavefronts of a workgroup are apart at the
barriers.

    python tools/stamp_aggregate.py build            # here (no GPU)
    python tools/stamp_aggregate.py run [S:copies]   # on the GPU box
"""
import ctypes, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ablate_aggregate as ab                                            # noqa: E402
PATCH = os.environ.get("STAMP_PATCH", "aggregate_stamps.patch")
TAG = os.environ.get("STAMP_TAG", "")                                     # a variant: STAMP_SUBS="old=>new@@old2=>new2" edits the patched copy
SUBS = [v.split("=>") for v in os.environ.get("STAMP_SUBS", "").split("@@") if v]
LIB = os.path.join(ab.OUT, "libgnnpn_hip_" + PATCH.replace(".patch", "") + TAG + ".so")
SLOTS, WAVES, WGS = 32, 16, 4096


def build():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gnnpn_build", os.path.join(ab.PKG, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()
    os.makedirs(ab.OUT, exist_ok=True)
    csrc = ab.patched_csrc(PATCH)
    src = open(os.path.join(csrc, "graph_tiled.hip")).read()
    for old, new in SUBS:
        assert src.count(old) == 1, old
        src = src.replace(old, new)
    open(os.path.join(csrc, "graph_tiled.hip"), "w").write(src)
    flags0 = [f for f in b.FLAGS if f != "-I" + b.CSRC] + ["-I" + csrc]
    o = os.path.join(ab.OUT, "graph_tiled_stamps" + TAG + ".o")
    subprocess.run(["hipcc"] + flags0 + ["-c", os.path.join(csrc, "graph_tiled.hip"), "-o", o], check=True)
    rest = [os.path.join(ab.PKG, "build", s.replace(".hip", ".o")) for s in b.SOURCES if s != "graph_tiled.hip"]
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + rest + [o], check=True)
    print("built", LIB)


def run(cfg, stagger=0, touch=0, planes=0):
    os.environ["GNNPN_LIB"] = LIB
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from gnnpn_sc_amd import _lib, graph, ops, synth
    S, copies = (int(v) for v in cfg.split(":"))
    dev = torch.device("cuda:0")
    table = synth.make_service_table(47, S, 0, degree=32, graph="scan")
    csr = graph.gcn_csr(torch.from_numpy(table.edge_index), torch.from_numpy(table.edge_attr), S)
    nnz = csr.col.numel()
    rp = torch.cat([csr.rowptr[:-1].long() + c * nnz for c in range(copies)] + [torch.tensor([copies * nnz])]).int().to(dev)
    col = torch.cat([csr.col.long() + c * S for c in range(copies)]).int().to(dev)
    w = csr.w.repeat(copies).to(dev)
    norm = ops.gcn_norm(rp, col, w)
    x = torch.randn(copies * S, 256, device=dev)
    bias = torch.randn(256, device=dev)
    scale, shift = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev)
    plan = ops.csr_tile_plan(rp, col, norm, S)
    assert plan.valid
    NT = plan.geom["src_tiles"]
    lib = _lib.load()
    assert lib.gnnpn_debug_agg_stagger(int(stagger)) == 0
    assert lib.gnnpn_debug_agg_touch(int(touch)) == 0
    assert lib.gnnpn_debug_agg_planes(0) == 0
    want = plan.aggregate(x, None, bias, scale, shift, ops.ACT_RELU)
    x_rows = x
    if planes:                                                            # [C/16][N][16]: a 16-channel slice's rows are contiguous
        x = x.view(-1, 16, 16).permute(1, 0, 2).contiguous().view(-1, 256)
        assert lib.gnnpn_debug_agg_planes(1) == 0
        assert torch.equal(plan.aggregate(x, None, bias, scale, shift, ops.ACT_RELU), want), "plane-major input: different result"
    rounds = []
    for rnd in range(4):
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        for _ in range(20):
            plan.aggregate(x, None, bias, scale, shift, ops.ACT_RELU)
        r1.record()
        torch.cuda.synchronize()
        rounds.append(round(r0.elapsed_time(r1) / 20, 4))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plan.aggregate(x, None, bias, scale, shift, ops.ACT_RELU)
    e1.record()
    torch.cuda.synchronize()
    buf = np.zeros(WGS * WAVES * SLOTS, np.uint64)
    lib.gnnpn_debug_agg_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    assert lib.gnnpn_debug_agg_stamps(buf.ctypes.data, buf.nbytes) == 0
    st = buf.reshape(WGS, WAVES, SLOTS).astype(np.int64)
    n = 5 + 3 * NT
    if os.environ.get("STAMP_DUMP"):
        np.save(os.path.join(os.environ["STAMP_DUMP"], f"stamps_{S}x{copies}_stg{stagger}_t{touch}_p{planes}.npy"), np.concatenate([st[:, :, :n].max(axis=1), st[:, :, SLOTS - 2:].max(axis=1)], axis=1))
    st = st[st[:, 0, n - 1] > 0][:, :, :n]                                 # workgroups that ran to the end
    t0 = st[:, :, 0].min(axis=1, keepdims=True)
    rel = st - t0[:, :, None]                                             # cycles since the workgroup's first wave entered
    names = ["entry", "headers requested"]
    for t in range(NT):
        names += [f"tile {t}: own fill written", f"tile {t}: barrier passed", f"tile {t}: own units done"]
    names += ["epilogue arithmetic done", "stores issued", "stores acknowledged"]
    rec = {"S": S, "copies": copies, "NT": NT, "variant": TAG, "stagger_cycles": stagger, "touch": touch, "planes": planes, "launch_ms": round(e0.elapsed_time(e1), 4), "rounds_ms": rounds[1:], "workgroups": int(st.shape[0]),
           "unit": "shader cycles since the workgroup's first entry", "stamps": []}
    for i, nm in enumerate(names):
        v = rel[:, :, i]
        rec["stamps"].append({"stamp": nm, "mean": round(float(v.mean()), 0), "first_wave": round(float(v.min(axis=1).mean()), 0),
                              "last_wave": round(float(v.max(axis=1).mean()), 0)})
    d = np.diff(rel, axis=2)
    rec["intervals_mean_cycles"] = {f"{names[i]} -> {names[i + 1]}": round(float(d[:, :, i].mean()), 0) for i in range(n - 1)}
    rec["workgroup_cycles_mean"] = round(float(rel[:, :, n - 1].max(axis=1).mean()), 0)
    # the spread of the wavefronts when they arrive at each tile's closing barrier (what the barrier costs the early ones)
    rec["arrival_spread_cycles"] = {f"tile {t}": round(float((rel[:, :, 4 + 3 * t].max(axis=1) - rel[:, :, 4 + 3 * t].min(axis=1)).mean()), 0)
                                    for t in range(NT)}
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        for c in (sys.argv[2:] or ["5000:128", "2507:256", "20000:8"]):
            for stg in [int(v) for v in os.environ.get("STAMP_STAGGER", "0").split(",")]:
                for tch in [int(v) for v in os.environ.get("STAMP_TOUCH", "0").split(",")]:
                    for pl in [int(v) for v in os.environ.get("STAMP_PLANES", "0").split(",")]:
                        run(c, stg, tch, pl)
