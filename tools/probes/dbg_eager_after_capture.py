"""After a PipelinedRunner's graph capture, run the eager single-stream pass and look at ITS status (default workspaces):
python tools/probes/dbg_eager_after_capture.py [workload] [trials] [precision]"""
import sys, os, json, time, torch
sys.path.insert(0, os.getcwd())
import gnnpn_sc_amd.synth as synth
from bench import WORKLOADS, build_models
from gnnpn_sc_amd import ops, _lib
from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
wl = sys.argv[1] if len(sys.argv) > 1 else "synth4"
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 4
prec = (sys.argv[3] if len(sys.argv) > 3 else None) or None
w = dict(WORKLOADS[wl])
T, K, S, B = w["T"], w["K"], w["S"], w["B"]
dev = torch.device("cuda:0")
table = synth.make_service_table(T, S, seed=0, degree=32)
svc = DeviceServices.from_table(table, dev)
batch = DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=4, tasks_per_problem=w["n_t"]), dev)
keys = ("idx_low", "idx_high", "R")


def area(buf):
    if buf is None:
        return None
    wds = buf[:2048].view(torch.int32).tolist()
    return {"words0_8": wds[:8], "seats_per_xcd": wds[256:264], "arrivals_per_xcd": wds[288:296]}


for trial in range(trials):
    net, low, high = build_models(T, S, K, dev, w["n_gcn"])
    pipe = ML2PNPipeline(net, low, high, K, precision=prec)
    runner = PipelinedRunner(pipe, svc, batch, slots=2, halves=False if wl != "qws" else None)
    dws = ops.workspaces(dev)
    recs = []
    for rep in range(3):
        t0 = time.perf_counter()
        out = pipe.run(svc, batch, decode_impl=runner.decode_impl)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        dump = {"encode": area(dws._encode), "decode": area(dws._decode)}
        st = dws.poll()
        rec = {"rep": rep, "ms": round(dt * 1e3, 2), "status": st, "staffing": int(_lib.load().gnnpn_coop_staffing_count())}
        if st:
            rec["ws"] = dump
            if st & 2:
                fr = ops.decode_failure_record()
                rec["decode_failures"] = {"failures": fr["failures"], "first": fr["records"][:2]}
        recs.append((rec, {k: out[k].clone() for k in keys}))
    same = [all(torch.equal(recs[i][1][k], recs[2][1][k]) for k in keys) for i in range(3)]
    for _ in range(6):
        runner.submit()
    rst = runner.poll()
    slot_same = [all(torch.equal(runner.graphs[s].outputs[k], recs[2][1][k]) for k in keys) for s in range(runner.n_slots)]
    print(json.dumps({"trial": trial, "eager": [r for r, _ in recs], "eager_equals_third": same, "runner_status": rst, "slots_equal_third_eager": slot_same}), flush=True)
    del runner, pipe, net, low, high, recs
    torch.cuda.synchronize()
