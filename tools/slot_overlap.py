"""Do the two slots of the pipeline really run side by side?  Submits `rounds` x 2 batches (slot 0, slot 1, ...) with HIP
events around every replay on the slot's stream and prints each replay's start/end relative to the round's start.
    python tools/slot_overlap.py [--workload synth5] [--batch B] [--rounds 6] [--precision split]"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_models
import gnnpn_sc_amd.synth as synth
from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="synth5")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--per-round", type=int, default=4)
ap.add_argument("--precision", default="split")
a = ap.parse_args()
w = dict(WORKLOADS[a.workload])
T, K, S, B = w["T"], w["K"], w["S"], a.batch or w["B"]
dev = torch.device("cuda:0")
table = synth.make_service_table(T, S, seed=0, degree=32)
pb = synth.make_problem_batch(table, B, seed=1, tasks_per_problem=T)
net, low, high = build_models(T, S, K, dev, w["n_gcn"])
pipe = ML2PNPipeline(net, low, high, K, precision=a.precision)
svc, batch = DeviceServices.from_table(table, dev), DeviceBatch.from_problems(pb, dev)
runner = PipelinedRunner(pipe, svc, batch, slots=2)
print(f"{a.workload} B={B} {a.precision}: halves={runner.halves} slots={runner.n_slots} lockstep={runner.lockstep}", flush=True)
for _ in range(2):
    runner.submit(batch)
runner.synchronize()
for r in range(a.rounds):
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t0.record()
    ev = []
    for i in range(a.per_round):
        s = runner.count % runner.n_slots
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(runner.stream(s)):
            if i < runner.n_slots:
                runner.stream(s).wait_event(t0)
            e0.record()
        runner.submit(batch)
        with torch.cuda.stream(runner.stream(s)):
            e1.record()
        ev.append((s, e0, e1))
    runner.synchronize()
    print(f"round {r}: " + "  ".join(f"slot{s} [{t0.elapsed_time(e0):8.3f} .. {t0.elapsed_time(e1):8.3f}]" for s, e0, e1 in ev), flush=True)
    # status area of each slot's LAST encoder / decoder launch: [status, workgroups on the same-XCD path, seats taken off the canonical CU, -]
    print("         " + "  ".join(f"ws{i}: enc {x.encode()[:16].view(torch.int32).tolist()} dec {x._decode[:16].view(torch.int32).tolist()}"
                                   for i, x in enumerate(runner.workspaces)), flush=True)
