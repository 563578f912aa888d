"""Free-running soak of the two-slot pipeline: rounds of `steps` submits that are NOT paced from the host (nothing is waited for inside
a round), each round started with a random delay (0..max_delay us, spin kernel) on a random slot so that the slots run through every
phase relation; after every round the sticky status words are polled and both slots' outputs compared with the single-stream run.
tools/soak_pipeline.py checks every step and thereby paces the slots from the host — the placement regression of round 6
(profiles/LOG_r06.md section 16) left no trace in it; here the old library shows the symptom within 150 rounds (78 declined and 22
off-canonical seats where the fixed library has none in 2000), though not yet a time-out: the fixed delays of
tools/probes/stagger_probe.py and tests/test_gpu_pipeline.py::test_two_slots_out_of_phase are what provokes those.
    python tools/soak_free_running.py [rounds=1000] [precision=split] [steps=100] [max_delay_us=800] [workload=qws]"""
import json, os, random, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_models
import gnnpn_sc_amd.synth as synth
from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
prec = sys.argv[2] if len(sys.argv) > 2 else "split"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
max_delay = int(sys.argv[4]) if len(sys.argv) > 4 else 800
wl = sys.argv[5] if len(sys.argv) > 5 else "qws"
w = dict(WORKLOADS[wl]); T, K, S, B = w["T"], w["K"], w["S"], w["B"]
dev = torch.device("cuda:0")
table = synth.make_service_table(T, S, seed=0, degree=32)
net, low, high = build_models(T, S, K, dev, w["n_gcn"])
pipe = ML2PNPipeline(net, low, high, K, precision=prec)
svc = DeviceServices.from_table(table, dev)
batch = DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=1, tasks_per_problem=w["n_t"]), dev)
runner = PipelinedRunner(pipe, svc, batch, slots=2, auto_degrade=False)
ref = pipe.run(svc, batch, decode_impl=runner.decode_impl)
for _ in range(8):
    runner.submit()
runner.synchronize(check=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)
rng = random.Random(2026)
bad_rounds, wrong, slow, ms_all, t_start = 0, 0, 0, [], time.time()
for r in range(rounds):
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for s in range(runner.n_streams):
        runner.stream(s).wait_event(t0)
    if runner.n_streams > 1:
        with torch.cuda.stream(runner.stream(rng.randrange(2))):
            torch.cuda._sleep(int(rng.uniform(0, max_delay) * cyc_per_us))
    for _ in range(steps):
        runner.submit()
    cur = torch.cuda.current_stream()
    for s in range(runner.n_streams):
        cur.wait_stream(runner.stream(s))
    t1.record()
    torch.cuda.synchronize()
    ms_all.append(t0.elapsed_time(t1))
    word = runner.poll()
    if word:
        bad_rounds += 1
        print(f"round {r}: status {word:#x}, {ms_all[-1]:.1f} ms, progress {runner.progress()}", flush=True)
    for s in range(runner.n_slots):
        o = runner.graphs[s].outputs
        if not (torch.equal(o["idx_high"], ref["idx_high"]) and torch.equal(o["R"], ref["R"])):
            wrong += 1
            print(f"round {r}: slot {s} differs from the single-stream run", flush=True)
med = sorted(ms_all)[len(ms_all) // 2]
slow = sum(1 for v in ms_all if v > 1.25 * med)
seats = [wsp.last_seats for wsp in runner.workspaces]
print(json.dumps({"what": "free-running soak (tools/soak_free_running.py)", "workload": wl, "batch": B, "precision": prec, "rounds": rounds, "steps_per_round": steps,
                  "max_start_delay_us": max_delay, "rounds_with_status": bad_rounds, "outputs_differing": wrong, "round_ms_median": round(med, 3),
                  "round_ms_max": round(max(ms_all), 3), "rounds_slower_than_1.25x_median": slow, "seats_cumulative": seats, "mode": {"halves": runner.halves, "lockstep": runner.lockstep},
                  "seconds": round(time.time() - t_start, 1)}))
sys.exit(1 if (bad_rounds or wrong) else 0)
