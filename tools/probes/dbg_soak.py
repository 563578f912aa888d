"""Fresh runner per trial at the QWS two-slot shape: how often does a runner produce outputs that differ from the eager run, or a
status?  python tools/probes/dbg_soak.py [trials] [precision] [write_through]"""
import sys, os, torch, json
sys.path.insert(0, os.getcwd())
import gnnpn_sc_amd.synth as synth
from bench import build_models
from gnnpn_sc_amd import ops
from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
prec = (sys.argv[2] if len(sys.argv) > 2 else None) or None
wt = None
mode = sys.argv[3] if len(sys.argv) > 3 else ""
dev = torch.device("cuda:0")
T, S, K, B, n_t = 47, 2507, 5, 256, 10
table = synth.make_service_table(T, S, seed=0, degree=32)
keys = ("idx_low", "idx_high", "R", "scores", "candidate_ids", "pn_inputs")
for trial in range(trials):
    net, low, high = build_models(T, S, K, dev)
    pipe = ML2PNPipeline(net, low, high, K, precision=prec)
    svc = DeviceServices.from_table(table, dev)
    batches = [DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=300 + i, tasks_per_problem=n_t), dev) for i in range(2)]
    runner = PipelinedRunner(pipe, svc, batches[0], slots=2, write_through=wt)
    packed = [runner.pack(b) for b in batches]
    if mode == "gc":
        import gc
        gc.collect(); torch.cuda.synchronize()
    refs = [pipe.run(svc, b, decode_impl=runner.decode_impl) for b in batches]
    dws = ops.workspaces(dev)
    torch.cuda.synchronize()
    def area(buf):
        if buf is None: return None
        w = buf[:2048].view(torch.int32).tolist()
        return {"words0_8": w[:8], "seats_per_xcd": w[256:264], "arrivals_per_xcd": w[288:296]}
    dump_default = {"encode": area(dws._encode), "decode": area(dws._decode)}
    st_default = dws.poll()
    if st_default:
        dump_default["decode_failures"] = {k: (v if k == "failures" else v[:2]) for k, v in ops.decode_failure_record().items()}
    pending, nbad, first = [], {0: 0, 1: 0}, None
    status = 0
    for i in range(120):
        out, s = runner.submit(packed[i % 2])
        ev = torch.cuda.Event(); ev.record(runner.stream(s))
        pending.append((i, i % 2, ev, out))
        if len(pending) == runner.n_slots:
            j, vj, evj, oj = pending.pop(0)
            evj.synchronize()
            d = {k: int((oj[k] != refs[vj][k]).sum().item()) for k in keys}
            if any(d.values()):
                nbad[vj] += 1
                first = first or (j, vj, d)
    torch.cuda.synchronize()
    dump_runner = [{"encode": area(w._encode), "decode": area(w._decode), "sticky": w.status.tolist()} for w in runner.workspaces]
    status = runner.poll()
    from gnnpn_sc_amd import _lib
    extra = {"staffing": int(_lib.load().gnnpn_coop_staffing_count()), "default_ws_status_after_first_eager": st_default, "default_ws": dump_default if st_default else None}
    if nbad[0] or nbad[1]:
        again = [pipe.run(svc, b, decode_impl=runner.decode_impl) for b in batches]
        extra["eager_again_equals_first_eager"] = [all(torch.equal(again[i][k], refs[i][k]) for k in keys) for i in range(2)]
        o0, s0 = runner.submit(packed[0]); runner.synchronize(check=False); a = {k: o0[k].clone() for k in keys}
        o1, s1 = runner.submit(packed[0]); runner.synchronize(check=False); b_ = {k: o1[k].clone() for k in keys}
        extra["slots"] = [s0, s1]
        extra["batch0_slotA_equals_slotB"] = all(torch.equal(a[k], b_[k]) for k in keys)
        extra["batch0_slotA_equals_eager_again"] = all(torch.equal(a[k], again[0][k]) for k in keys)
        extra["batch0_slotB_equals_eager_again"] = all(torch.equal(b_[k], again[0][k]) for k in keys)
        extra["status_after"] = runner.poll()
    print(json.dumps({"trial": trial, "bad": nbad, "first": first, "status": status, "runner_ws": dump_runner if (status or nbad[0] or nbad[1]) else None, **extra}), flush=True)
    del runner, pipe, net, low, high, svc, batches, packed, refs
    torch.cuda.synchronize()
