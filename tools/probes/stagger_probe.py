"""Does a 20-step round of the two free-running slots lose time because both slots START together?  (The long run settles at
0.55 ms per step at QWS, 20-step rounds between synchronisations at 0.58; slots that are forced to start every step together —
lockstep — measured 281-350 k problems/s against 450 k.)  Times rounds of `--steps` submits with slot 1's FIRST replay of every
round held back by a spin kernel of d microseconds on its stream, for several d, and prints per-step times of the round too.
    python tools/probes/stagger_probe.py [--steps 20] [--rounds 30] [--delays 0,150,300,450,600]"""
import argparse, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import WORKLOADS, build_models
import gnnpn_sc_amd.synth as synth
from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="qws")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=30)
ap.add_argument("--delays", default="0,-150,-250,150,600")
ap.add_argument("--precision", default="split")
a = ap.parse_args()
w = dict(WORKLOADS[a.workload])
T, K, S, B = w["T"], w["K"], w["S"], w["B"]
dev = torch.device("cuda:0")
table = synth.make_service_table(T, S, seed=0, degree=32)
pb = synth.make_problem_batch(table, B, seed=1, tasks_per_problem=T)
net, low, high = build_models(T, S, K, dev, w["n_gcn"])
pipe = ML2PNPipeline(net, low, high, K, precision=a.precision)
svc, batch = DeviceServices.from_table(table, dev), DeviceBatch.from_problems(pb, dev)
runner = PipelinedRunner(pipe, svc, batch, slots=2, auto_degrade=False)      # a failed hand-off is reported per round, the form is not switched
assert runner.n_slots == 2 and not runner.lockstep
for _ in range(8):
    runner.submit(batch)
runner.synchronize(check=True)

# cycles of torch.cuda._sleep per microsecond, measured
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
torch.cuda._sleep(10_000_000)
e1.record()
torch.cuda.synchronize()
cyc_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)
print(f"{a.workload} B={B}: spin kernel {cyc_per_us:.1f} cycles/us", flush=True)


gate_stream = torch.cuda.Stream()


def one_round(delay_us, extra_us=0):
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    if delay_us < 0:                                   # negative: BOTH slots held behind one gate that opens |d| us after t0 — an exactly common start
        gate = torch.cuda.Event()
        with torch.cuda.stream(gate_stream):
            gate_stream.wait_event(t0)
            torch.cuda._sleep(int(-delay_us * cyc_per_us))
            gate.record()
        for s in range(2):
            runner.stream(s).wait_event(gate)
    else:
        for s in range(2):
            runner.stream(s).wait_event(t0)
    if delay_us > 0:
        with torch.cuda.stream(runner.stream(1)):
            torch.cuda._sleep(int(delay_us * cyc_per_us))
    if extra_us > 0:                                    # behind the common gate: slot 1 a little later than slot 0
        with torch.cuda.stream(runner.stream(1)):
            torch.cuda._sleep(int(extra_us * cyc_per_us))
    for _ in range(a.steps):
        runner.submit()
    cur = torch.cuda.current_stream()
    for s in range(2):
        cur.wait_stream(runner.stream(s))
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1)


for spec in a.delays.split(",") * 2:
    d, _, x = spec.partition(":")                      # "d" or "d:x": x = extra microseconds for slot 1 behind a common gate (d < 0)
    d, x = int(d), int(x or 0)
    one_round(d, x)
    ms, words = [], []
    for _ in range(a.rounds):
        ms.append(one_round(d, x))
        words.append(runner.poll())                     # the slots' sticky status words, OR-ed (0: every launch of the round did its work)
    if any(words):
        print(f"delay {d:4d} us: status words per round {[hex(v) for v in words]}; progress {runner.progress()}", flush=True)
    med = statistics.median(ms)
    print(f"delay {d:4d} us extra {x:3d}: round median {med:.3f} ms  min {min(ms):.3f}  p90 {sorted(ms)[int(0.9 * len(ms))]:.3f}  "
          f"slow rounds (> 1.02 x median) {sum(1 for v in ms if v > 1.02 * med)} of {len(ms)}  -> {B * a.steps / med:.0f} k problems/s", flush=True)
try:
    runner.synchronize(check=True)
except Exception as err:                              # report, then fail
    from gnnpn_sc_amd import ops
    print("FAILED:", str(err)[:600], flush=True)
    print("failure record:", ops.decode_failure_record(clear=False), flush=True)
    sys.exit(1)
