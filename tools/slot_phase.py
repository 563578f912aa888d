"""Phase offset between the two slots' loops (end-of-step events), in the fast and the slow attractor."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnpn_sc_amd import synth
from gnnpn_sc_amd.pipeline import ML2PNPipeline, DeviceServices, DeviceBatch, PipelinedRunner
import bench
dev = torch.device("cuda:0")
w = bench.WORKLOADS["qws"]
table = synth.make_service_table(w["T"], w["S"], seed=0, degree=32)
pb = synth.make_problem_batch(table, w["B"], seed=1, tasks_per_problem=w["n_t"])
net, low, high = bench.build_models(w["T"], w["S"], w["K"], dev, w["n_gcn"])
pipe = ML2PNPipeline(net, low, high, w["K"])
svc, batch = DeviceServices.from_table(table, dev), DeviceBatch.from_problems(pb, dev)
runner = PipelinedRunner(pipe, svc, batch, slots=2)
import gc; gc.disable()
def lockstep(n):
    ev = [torch.cuda.Event(), torch.cuda.Event()]
    for i in range(n):
        runner.submit(); ev[0].record(runner.stream(0))
        runner.submit(); ev[1].record(runner.stream(1))
        runner.stream(0).wait_event(ev[1]); runner.stream(1).wait_event(ev[0])
def free(n, label):
    base = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    base.record()
    evs = [[], []]
    t0 = time.perf_counter()
    for i in range(n):
        for s in (0, 1):
            runner.submit()
            e = torch.cuda.Event(enable_timing=True); e.record(runner.stream(s)); evs[s].append(e)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = [[base.elapsed_time(e) for e in evs[s]] for s in (0, 1)]
    half = n // 2
    per = [(t[s][-1] - t[s][half]) / (n - 1 - half) for s in (0, 1)]
    off = [(t[1][i] - t[0][i]) for i in range(half, n)]
    print(f"{label}: {1e3 * dt / (2 * n):.4f} ms/step; slot periods {per[0]:.3f} {per[1]:.3f} ms; "
          f"slot1 ends after slot0 by {sum(off) / len(off):.3f} ms (min {min(off):.3f} max {max(off):.3f})")
from gnnpn_sc_amd import ops
def fast_path_counts():
    out = []
    for slot in (0, 1):
        ops.set_workspace_slot(slot)
        ws = ops.encode_workspace(dev)
        wd = ops.decode_workspace(dev, w["B"], w["T"], w["K"])
        out.append((int(ws[4:8].view(torch.int32).item()), int(wd[4:8].view(torch.int32).item())))
    ops.set_workspace_slot(0)
    return out

def quarters(n, label):
    torch.cuda.synchronize()
    ts, fp = [], []
    t0 = time.perf_counter()
    for q in range(4):
        for i in range(n // 4):
            runner.submit(); runner.submit()
        torch.cuda.synchronize()
        ts.append(time.perf_counter())
        fp.append(fast_path_counts())
    print(label, "ms/step per quarter:", [round(1e3 * (b - a) / (2 * (n // 4)), 4) for a, b in zip([t0] + ts[:-1], ts)],
          "fast-path WGs (encoder, decoder) of the last launch per slot, per quarter:", fp)
quarters(400, "fresh")
for pause in (0.0, 0.1, 0.3, 1.0, 3.0, 0.1, 0.3, 1.0):
    time.sleep(pause)
    quarters(400, f"after {pause:.1f} s idle")
