"""Soak test of the two-slot pipeline: N steps with a different batch every step, every output compared with a
single-stream run of the same batch (bit for bit), hand-off status checked at the end.
Usage: soak_pipeline.py [N] [precision] [workload] [batch]   (batches of 512 and more exercise the half-batch pairing)
Environment: SOAK_POLL_EVERY=n  the runner's poll() every n steps (default 500): status 0 and the proof of work — workgroup-tiles
                                finished == expected == the host's count — at EVERY poll, or the soak fails;
             SOAK_JITTER_US=n   before every submit a spin kernel of 0..n us (seeded) on that slot's stream: the two slots drift through
                                every phase relation — cooperative launches beside the other slot's ordinary kernels, beside each
                                other, staffing at the same moment (round 6: the placement regression of LOG_r06 section 16 only showed
                                out of phase);
             SOAK_DIST=1        a real RCCL process group of world size 1 and the path's single collective, the asynchronous
                                all-gather of the selected indices, behind every step on the group's own stream (what a rank of
                                `bench.py --gpus N` does)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import ops, synth
from gnnpn_sc_amd.pipeline import ML2PNPipeline, DeviceServices, DeviceBatch, PipelinedRunner
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda:0")
w = dict(bench.WORKLOADS[sys.argv[3] if len(sys.argv) > 3 else "qws"])
if len(sys.argv) > 4:
    w["B"] = int(sys.argv[4])
table = synth.make_service_table(w["T"], w["S"], seed=0, degree=32)
net, low, high = bench.build_models(w["T"], w["S"], w["K"], dev, w["n_gcn"])
pipe = ML2PNPipeline(net, low, high, w["K"], precision=prec)
svc = DeviceServices.from_table(table, dev)
n_var = 12
batches = [DeviceBatch.from_problems(synth.make_problem_batch(table, w["B"], seed=100 + i, tasks_per_problem=w["n_t"]), dev)
           for i in range(n_var)]
poll_every = int(os.environ.get("SOAK_POLL_EVERY", 500))
use_dist = os.environ.get("SOAK_DIST") == "1"
if use_dist:                                          # before the runner: beside a process group its slots' streams take a priority of their own
    from gnnpn_sc_amd import dist as gdist
    gdist.init_process_group("nccl", dev)
runner = PipelinedRunner(pipe, svc, batches[0], slots=2, auto_degrade=False)
gathers, polls, bad_polls = {}, 0, 0
refs = [pipe.run(svc, b, decode_impl=runner.decode_impl) for b in batches]           # single stream, same kernels
torch.cuda.synchronize()
keys = ("idx_low", "idx_high", "R", "actions", "win_low", "win_high_raw")
bad, t0 = 0, time.time()
pending = []                                          # (step, slot, variant, event)
jitter_us = int(os.environ.get("SOAK_JITTER_US", 0))
if jitter_us:
    import random
    rng = random.Random(12345)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
    cyc_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)
for i in range(N):
    v = (i * 7 + i // 5) % n_var
    if jitter_us and rng.random() < 0.5:
        with torch.cuda.stream(runner.stream(runner.count % runner.n_slots)):
            torch.cuda._sleep(int(rng.uniform(0, jitter_us) * cyc_per_us))
    if use_dist and gathers.get(runner.count % runner.n_slots, (None, None))[1] is not None:
        with torch.cuda.stream(runner.stream(runner.count % runner.n_slots)):
            gathers[runner.count % runner.n_slots][1].wait()          # the previous gather out of this slot's index buffer is done
    out, s = runner.submit(batches[v])
    if use_dist:
        with torch.cuda.stream(runner.stream(s)):
            gathers[s] = gdist.all_gather_indices_async(out["idx_high"], gathers.get(s, (None, None))[0])
    ev = torch.cuda.Event(); ev.record(runner.stream(s))
    snap = None
    pending.append((i, s, v, ev, out))
    if len(pending) == runner.n_slots:                # check the oldest step before its slot is reused
        j, sj, vj, evj, oj = pending.pop(0)
        evj.synchronize()
        for k in keys:
            if not torch.equal(oj[k], refs[vj][k]):
                bad += 1
                wsj = runner.workspaces[sj if not runner.halves else 0]
                print(f"step {j} slot {sj} batch {vj}: {k} differs ({int((oj[k] != refs[vj][k]).sum())} elements); status areas of the slot's last launches "
                      f"[error word, same-XCD workgroups, off-canonical seats, -]: enc {wsj.encode()[:16].view(torch.int32).tolist()} "
                      f"dec {wsj._decode[:16].view(torch.int32).tolist()}; sticky {[int(x.status[0]) for x in runner.workspaces]}", flush=True)
                break
    if poll_every and (i + 1) % poll_every == 0:
        for g_, w_ in gathers.values():
            if w_ is not None:
                w_.wait()
        word = runner.poll()
        polls += 1
        prog = runner.progress()
        ok = word == 0 and all(p[part]["finished"] == p[part]["expected"] == p[part]["host_expected"] for p in prog for part in ("encoder", "decoder"))
        if not ok:
            bad_polls += 1
            print(f"poll after step {i}: status {word:#x}, progress {prog}", flush=True)
        for j, sj, vj, evj, oj in pending:            # everything submitted so far has finished: compare it before the slots are reused
            for k in keys:
                if not torch.equal(oj[k], refs[vj][k]):
                    bad += 1; print(f"step {j}: {k} differs", flush=True); break
        pending = []
for j, sj, vj, evj, oj in pending:
    evj.synchronize()
    for k in keys:
        if not torch.equal(oj[k], refs[vj][k]):
            bad += 1; print(f"step {j}: {k} differs"); break
if bad:
    print("failure record (first entries):", ops.decode_failure_record(clear=False))
ops.check_status(dev)
import json
print(json.dumps({"steps": N, "precision": prec, "workload": sys.argv[3] if len(sys.argv) > 3 else "qws", "batch": w["B"], "rccl_world1_all_gather_per_step": use_dist,
                  "mismatching_steps": bad, "polls": polls, "polls_with_status_or_shortfall": bad_polls, "progress_at_end": runner.progress() if polls else None,
                  "write_through": runner.write_through, "front_lds_kb": runner.front_lds_kb, "slot_stream_priority": runner.stream_priority, "jitter_us": jitter_us, "seconds": round(time.time() - t0, 1)}))
if use_dist:
    gdist.destroy(2)
sys.exit(1 if (bad or bad_polls) else 0)
