"""The workflow (GIN) branch at the 1000-task shapes: one launch per GIN layer (gnnpn_gin_layer_f32, nodeLin behind the last)
against the layered path (csr_aggregate + linear + linear per layer, nodeLin), same bits; and the one-launch layer on the fp16
matrix cores through the exact split (gnnpn_gin_layer_split).
    python tools/bench_front_half.py [--workload synth4] [--batch 512]"""
import argparse, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_models
import gnnpn_sc_amd.synth as synth
from gnnpn_sc_amd.pipeline import DeviceBatch

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="synth4")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
w = dict(WORKLOADS[a.workload])
T, K, S, B = w["T"], w["K"], w["S"], a.batch or w["B"]
dev = torch.device("cuda:0")
table = synth.make_service_table(T, S, seed=0, degree=32)
net, low, high = build_models(T, S, K, dev, w["n_gcn"])
batch = DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=1, tasks_per_problem=w["n_t"]), dev)


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


run = lambda prec="f32": net.request_embedding(batch.x, batch.wf_csr, batch.seg_ptr, batch.max_nodes, prec)   # noqa: E731
net.fuse_gin_layers = False
want = run()
ms_layered = timed(run)
net.fuse_gin_layers = True
got = run()
ms_fused = timed(run)
split = run("split")
ms_split = timed(lambda: run("split"))
rel = ((split.double() - got.double()).abs().amax() / got.double().abs().amax()).item()
rows = batch.x.shape[0]
flop = rows * 2 * (26 * 256 + 256 * 128 + 128 * 256 + 256 * 128 + 128 * 128)
print(json.dumps({"workload": a.workload, "problems": B, "rows": rows, "layered_ms": round(ms_layered, 4), "fused_ms": round(ms_fused, 4),
                  "fused_split_ms": round(ms_split, 4), "split_vs_f32_max_rel_diff": rel,
                  "bit_identical": bool(torch.equal(got, want)), "dense_GFLOP": round(flop / 1e9, 1),
                  "fused_TFLOPs_incl_aggregate_and_mean": round(flop / ms_fused / 1e9, 1)}))
