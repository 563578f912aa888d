"""HBM roofline of the CSR gather-aggregate on the reference's own batching of the service graph: a PyG batch of B problems
holds B copies of the service table (trainML.py:109-114, modelML.py:145-156), i.e. one GCN aggregate over B*S rows per
layer.  Times both forms on the same operands and checks that they agree bit for bit:
  gather : gnnpn_csr_aggregate_f32        (one wave per destination row, source rows gathered from L2)
  lds    : gnnpn_csr_aggregate_blocks_f32 (north star "node features staged in LDS": one workgroup per (copy, channel slice))
  tiled  : gnnpn_csr_aggregate_tiled_f32  (destination tile x source tile, one source tile's slice in LDS at a time, the edge
           lists from the plan's sliced-ELL stream; needs rows in source order: --graph scan, the reference's emission order)

    python tools/bench_aggregate.py [--configs S:copies,...] [--degree 32] [--graph scan|random]

Algorithmic bytes per launch: 2*N*C*4 + E*(4+4) + (N+1)*4, N = copies*S, C = 256.
"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import _lib, graph, ops, synth
from gnnpn_sc_amd._lib import check, dev_ptr, stream_ptr

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="2507:256,2507:64,5000:128,5000:32,10000:32,20000:8")
ap.add_argument("--degree", type=int, default=32)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--graph", default="scan", help="scan: pairs in the reference's emission order (rows sorted by source); random: as drawn")
ap.add_argument("--forms", default="gather,lds,tiled", help="gather, lds, tiled")
ap.add_argument("--no-check", action="store_true", help="timing-only (ablation) builds: do not compare the forms")
ap.add_argument("--tiled-forms", default="default", help="only with tools/experiments/aggregate_two_buffers.patch applied: 0 (one tile buffer), 1 (two), or 0,1")
a = ap.parse_args()
dev = torch.device("cuda:0")
F32, I32 = torch.float32, torch.int32


def blocks_form(rp, col, w, x, bias, scale, shift, S, order=None):
    n, C = x.shape
    y = torch.empty_like(x)
    check(_lib.load().gnnpn_csr_aggregate_blocks_f32(
        dev_ptr(rp, I32, "rowptr"), dev_ptr(col, I32, "col"), dev_ptr(w, F32, "w"), dev_ptr(x, F32, "x"), C, None,
        dev_ptr(bias, F32, "b"), dev_ptr(scale, F32, "s"), dev_ptr(shift, F32, "t"), ops.ACT_RELU, dev_ptr(y, F32, "y"), C, n, C,
        S, dev_ptr(order, I32, "order", True), stream_ptr()), "gnnpn_csr_aggregate_blocks_f32")
    return y


def timed(fn, reps):
    """MEDIAN of 5 rounds of ``reps`` back-to-back launches after a warm-up round of the same length (the first rounds after a
    change of kernel run at a lower clock: whichever form was timed first used to look 10 % slower).  timed.rounds keeps the
    rounds, timed.best their minimum — rounds 1-5 quoted the minimum; profiles/LOG_r06.md section 1 compares both with rocprofv3's
    per-dispatch durations of the same launches (they agree within 1-2 % once the clock has settled)."""
    rounds = []
    for rnd in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            rounds.append(e0.elapsed_time(e1) / reps)
    timed.rounds = [round(v, 4) for v in rounds]
    timed.best = min(rounds)
    return sorted(rounds)[len(rounds) // 2]


out = []
for cfg in a.configs.split(","):
    S, copies = (int(v) for v in cfg.split(":"))
    table = synth.make_service_table(47, S, 0, degree=a.degree, graph=a.graph)
    csr = graph.gcn_csr(torch.from_numpy(table.edge_index), torch.from_numpy(table.edge_attr), S)
    nnz = csr.col.numel()
    rp = torch.cat([csr.rowptr[:-1].long() + c * nnz for c in range(copies)] + [torch.tensor([copies * nnz])]).int().to(dev)
    col = torch.cat([csr.col.long() + c * S for c in range(copies)]).int().to(dev)
    w = csr.w.repeat(copies).to(dev)
    N, C = copies * S, 256
    norm = ops.gcn_norm(rp, col, w)
    g = torch.Generator(device=dev).manual_seed(S + copies)
    x = torch.randn(N, C, device=dev, generator=g)
    bias = torch.randn(C, device=dev, generator=g)
    scale, shift = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g)
    ops.PREFER_LDS_AGGREGATE = False
    ops.PREFER_TILED_AGGREGATE = False
    gather = lambda: ops.csr_aggregate(rp, col, norm, x, bias=bias, scale=scale, shift=shift, act=ops.ACT_RELU)   # noqa: E731
    alg = 2 * N * C * 4 + nnz * copies * 8 + (N + 1) * 4
    rec = {"S": S, "copies": copies, "rows": N, "channels": C, "nnz": nnz * copies, "algorithmic_bytes": alg, "graph": a.graph}
    forms = a.forms.split(",")
    want = gather()
    if "gather" in forms:
        ms_g = timed(gather, a.reps)
        rec["gather"] = {"ms": round(ms_g, 4), "GBps": round(alg / ms_g / 1e6, 1), "frac_of_8TBps": round(alg / ms_g / 8e9, 4)}
    if "tiled" in forms:
        tforms = [None] if a.tiled_forms == "default" else [int(v) for v in a.tiled_forms.split(",")]
        for tf in tforms:
            if tf == 1 and S > getattr(ops, "TILED2_ROWS_MAX", 0):
                continue
            plan = ops.csr_tile_plan(rp, col, norm, S) if tf is None else ops.TilePlan(rp, col, norm, S, form=tf)
            st = plan.stats
            key = "tiled" if len(tforms) == 1 else f"tiled_form{tf}"
            rec[key] = {"plan_valid": plan.valid, **({"geometry": plan.geom} if plan.stats else {}),
                        "stream": {k: st.get(k) for k in ("edges", "slots", "efficiency", "stream_bytes", "mean_run")},
                        "run_histogram": st.get("run_histogram")}
            if plan.valid:
                tiled = lambda: plan.aggregate(x, None, bias, scale, shift, ops.ACT_RELU)   # noqa: E731
                assert a.no_check or torch.equal(tiled(), want), f"tiled form {tf} differs from the gather form at S={S}"
                ms_t = timed(tiled, a.reps)
                alg_t = 2 * N * C * 4 + st["stream_bytes"] + N * 8          # what this form reads instead of the CSR: stream, row order, self-loop weights
                rec[key].update(ms=round(ms_t, 4), ms_median=round(ms_t, 4), ms_best_round=round(timed.best, 4), rounds_ms=timed.rounds, GBps=round(alg / ms_t / 1e6, 1),
                                frac_of_8TBps=round(alg / ms_t / 8e9, 4), bytes_this_form_moves=alg_t, bit_identical_to_gather=not a.no_check)
            del plan
    if S <= ops.LDS_BLOCK_ROWS_MAX and "lds" in forms:
        order = ops.csr_block_row_order(rp, S)                              # once per graph
        lds = lambda: blocks_form(rp, col, norm, x, bias, scale, shift, S, order)   # noqa: E731
        lds_unordered = lambda: blocks_form(rp, col, norm, x, bias, scale, shift, S)   # noqa: E731
        assert torch.equal(lds(), want), f"LDS-staged form differs from the gather form at S={S}"
        assert torch.equal(lds_unordered(), want), f"LDS-staged form (rows as they come) differs from the gather form at S={S}"
        ms_l, ms_u = timed(lds, a.reps), timed(lds_unordered, a.reps)
        rec["lds"] = {"ms": round(ms_l, 4), "GBps": round(alg / ms_l / 1e6, 1), "frac_of_8TBps": round(alg / ms_l / 8e9, 4),
                      "ms_rows_as_they_come": round(ms_u, 4),
                      "slice_channels": next(4 * c for c in (4, 2, 1) if (S + 1) * 16 * c <= 160 * 1024), "bit_identical_to_gather": True}
    out.append(rec)
    print(json.dumps(rec), flush=True)
    del rp, col, w, x, norm
    torch.cuda.empty_cache()
