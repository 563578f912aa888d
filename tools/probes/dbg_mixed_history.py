"""Process history like the test suite's: runners of different shapes captured one after another, an eager pass right after each
capture, default-workspace status read after every eager pass.  python tools/probes/dbg_mixed_history.py [rounds]"""
import sys, os, json, time, torch
sys.path.insert(0, os.getcwd())
import gnnpn_sc_amd.synth as synth
from bench import WORKLOADS, build_models
from gnnpn_sc_amd import ops, _lib
from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
keys = ("idx_low", "idx_high", "R")
ctx = {}
for wl in ("qws", "synth4", "normal"):
    w = dict(WORKLOADS[wl])
    table = synth.make_service_table(w["T"], w["S"], seed=0, degree=32)
    ctx[wl] = (w, DeviceServices.from_table(table, dev),
               DeviceBatch.from_problems(synth.make_problem_batch(table, w["B"], seed=4, tasks_per_problem=w["n_t"]), dev))
n = 0
for rnd in range(rounds):
    for wl, halves in (("qws", None), ("synth4", False), ("normal", None), ("synth4", None), ("qws", None)):
        w, svc, batch = ctx[wl]
        net, low, high = build_models(w["T"], w["S"], w["K"], dev, w["n_gcn"])
        pipe = ML2PNPipeline(net, low, high, w["K"])
        runner = PipelinedRunner(pipe, svc, batch, slots=2, halves=halves)
        dws = ops.workspaces(dev)
        outs, sts = [], []
        for rep in range(2):
            out = pipe.run(svc, batch, decode_impl=runner.decode_impl)
            torch.cuda.synchronize()
            words = [int(b[:4].view(torch.int32).item()) if b is not None else None for b in (dws._encode, dws._decode)]
            st = dws.poll()
            rec = {"status": st, "launch_words": words}
            if st & 2:
                fr = ops.decode_failure_record()
                rec["decode_failures"] = {"failures": fr["failures"], "first": fr["records"][:1]}
            if st:
                wds = dws._encode[:2048].view(torch.int32).tolist()
                rec["encode_seats_per_xcd"], rec["encode_arrivals_per_xcd"] = wds[256:264], wds[288:296]
            sts.append(rec)
            outs.append({k: out[k].clone() for k in keys})
        for _ in range(8):
            runner.submit()
        rst = runner.poll()
        same01 = all(torch.equal(outs[0][k], outs[1][k]) for k in keys)
        slot_same = [all(torch.equal(runner.graphs[s].outputs[k], outs[1][k]) for k in keys) for s in range(runner.n_slots)]
        n += 1
        bad = (not same01) or rst or any(s["status"] for s in sts) or not all(slot_same)
        print(json.dumps({"n": n, "wl": wl, "halves": runner.halves, "eager": sts, "eager0_equals_eager1": same01, "runner_status": rst,
                          "slots_equal_eager1": slot_same, "BAD": bool(bad)}), flush=True)
        del runner, pipe, net, low, high
        torch.cuda.synchronize()
