"""Two slots whose replays start a few ms apart (what a host that consumes every result between submissions produces):
per step the encoder / decoder status areas [error word, same-XCD workgroups, off-canonical seats, -] and whether the
result equals the single-stream run.   python tools/repro_stagger.py [--workload synth5] [--steps 40] [--delay-ms 3]"""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_models
import gnnpn_sc_amd.synth as synth
from gnnpn_sc_amd import ops
from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="synth5")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--delay-ms", type=float, default=3.0)
ap.add_argument("--precision", default="split")
a = ap.parse_args()
w = dict(WORKLOADS[a.workload])
T, K, S, B = w["T"], w["K"], w["S"], w["B"]
dev = torch.device("cuda:0")
table = synth.make_service_table(T, S, seed=0, degree=32)
net, low, high = build_models(T, S, K, dev, w["n_gcn"])
pipe = ML2PNPipeline(net, low, high, K, precision=a.precision)
svc = DeviceServices.from_table(table, dev)
batches = [DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=100 + i, tasks_per_problem=w["n_t"]), dev) for i in range(3)]
runner = PipelinedRunner(pipe, svc, batches[0], slots=2)
refs = [pipe.run(svc, b, decode_impl=runner.decode_impl) for b in batches]
torch.cuda.synchronize()
print(f"{a.workload} B={B}: slots={runner.n_slots} halves={runner.halves} lockstep={runner.lockstep} delay {a.delay_ms} ms", flush=True)
prev = None
for i in range(a.steps):
    v = i % 3
    out, s = runner.submit(batches[v])
    ev = torch.cuda.Event(); ev.record(runner.stream(s))
    if prev is not None:
        j, sj, vj, evj, oj = prev
        evj.synchronize()
        ok = all(torch.equal(oj[k], refs[vj][k]) for k in ("idx_low", "idx_high", "R"))
        wsj = runner.workspaces[sj]
        print(f"step {j} slot {sj}: {'ok ' if ok else 'BAD'} enc {wsj.encode()[:16].view(torch.int32).tolist()} dec {wsj._decode[:16].view(torch.int32).tolist()} "
              f"sticky {[int(x.status[0]) for x in runner.workspaces]}", flush=True)
        time.sleep(a.delay_ms / 1e3)
    prev = (i, s, v, ev, out)
runner.synchronize(check=False)
print("failure record:", ops.decode_failure_record(clear=False) if hasattr(ops, "decode_failure_record") else None)
