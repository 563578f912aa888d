"""What does the status area of a cooperative launch look like when a staggered start of the two slots ends in a hand-off time-out?
Short rounds (a few steps), slot 1's first replay held back by d us; after every round the slots' sticky words are polled and, when
one is set, the status areas of both slots' LAST encoder and decoder launches are printed per XCD: seats taken, arrivals, members that
ran the L2-resident hand-off, seat flags, CUs reached.      python tools/probes/dbg_stagger_state.py [--delay 400] [--steps 6] [--rounds 200]"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import WORKLOADS, build_models
import gnnpn_sc_amd.synth as synth
from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner

ap = argparse.ArgumentParser()
ap.add_argument("--delay", type=int, default=400)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--rounds", type=int, default=200)
ap.add_argument("--dumps", type=int, default=3)
a = ap.parse_args()
w = dict(WORKLOADS["qws"])
T, K, S, B = w["T"], w["K"], w["S"], w["B"]
dev = torch.device("cuda:0")
table = synth.make_service_table(T, S, seed=0, degree=32)
pb = synth.make_problem_batch(table, B, seed=1, tasks_per_problem=T)
net, low, high = build_models(T, S, K, dev, w["n_gcn"])
pipe = ML2PNPipeline(net, low, high, K, precision="split")
svc, batch = DeviceServices.from_table(table, dev), DeviceBatch.from_problems(pb, dev)
runner = PipelinedRunner(pipe, svc, batch, slots=2, auto_degrade=False)
for _ in range(8):
    runner.submit(batch)
runner.synchronize(check=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
torch.cuda._sleep(10_000_000)
e1.record()
torch.cuda.synchronize()
cyc_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)


def area(buf):
    v = buf.view(torch.int32).cpu()
    per = lambda off: [int(v[off // 4 + 32 * x]) for x in range(8)]   # noqa: E731
    taken = [int(v[10240 // 4 + 64 * x: 10240 // 4 + 64 * x + 64].sum()) for x in range(8)]
    claims = v[2048 // 4: 2048 // 4 + 8 * 256].view(8, 256)
    return {"words0_7": [int(x) for x in v[:8]], "seats_taken": per(12288), "arrivals": per(13312), "l2_members": per(14336), "seat_flags": taken,
            "cus_reached": [int((claims[x] > 0).sum()) for x in range(8)], "max_claims_on_a_cu": [int(claims[x].max()) for x in range(8)]}


dumps = 0
for r in range(a.rounds):
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for s in range(2):
        runner.stream(s).wait_event(t0)
    with torch.cuda.stream(runner.stream(1)):
        torch.cuda._sleep(int(a.delay * cyc_per_us))
    for _ in range(a.steps):
        runner.submit()
    cur = torch.cuda.current_stream()
    for s in range(2):
        cur.wait_stream(runner.stream(s))
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1)
    stick = [[int(x) for x in wsp.status[:8].cpu()] for wsp in runner.workspaces]
    word = runner.poll()
    if word or ms > 50:
        print(f"round {r}: {ms:.1f} ms, status {word:#x}; sticky blocks {stick}", flush=True)
        for i, wsp in enumerate(runner.workspaces):
            print(f"  slot {i} encoder: {area(wsp._encode)}", flush=True)
            print(f"  slot {i} decoder: {area(wsp._decode)}", flush=True)
        dumps += 1
        if dumps >= a.dumps:
            break
print(f"{r + 1} rounds, {dumps} with a status", flush=True)
