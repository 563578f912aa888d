"""Soak test of the two-slot pipeline: N steps with a different batch every step, every output compared with a
single-stream run of the same batch (bit for bit), hand-off status checked at the end.
Usage: soak_pipeline.py [N] [precision] [workload] [batch]   (batches of 512 and more exercise the half-batch pairing)"""
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
runner = PipelinedRunner(pipe, svc, batches[0], slots=2)
refs = [pipe.run(svc, b, decode_impl=runner.decode_impl) for b in batches]           # single stream, same kernels
torch.cuda.synchronize()
keys = ("idx_low", "idx_high", "R", "actions", "win_low", "win_high_raw")
bad, t0 = 0, time.time()
pending = []                                          # (step, slot, variant, event)
for i in range(N):
    v = (i * 7 + i // 5) % n_var
    out, s = runner.submit(batches[v])
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
for j, sj, vj, evj, oj in pending:
    evj.synchronize()
    for k in keys:
        if not torch.equal(oj[k], refs[vj][k]):
            bad += 1; print(f"step {j}: {k} differs"); break
if bad:
    print("failure record (first entries):", ops.decode_failure_record(clear=False))
ops.check_status(dev)
print(f"{N} steps ({prec}), {bad} mismatching, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
