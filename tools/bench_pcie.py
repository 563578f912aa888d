"""The PCIe-inclusive rate of the path (never `value` of bench.py, whose inputs are resident in HBM): every step's problem
batch starts in pinned HOST memory (what a caller of the reference's Python API holds), is copied into the slot's static
inputs on the slot's stream, and the step's results (idx_high [B,T] int32, R [B]) are copied back to pinned host memory.
    python tools/bench_pcie.py [--workload qws] [--steps 400]"""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_models
import gnnpn_sc_amd.synth as synth
from gnnpn_sc_amd import graph
from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="qws")
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--precision", default="split")
a = ap.parse_args()
w = dict(WORKLOADS[a.workload])
T, K, S, B = w["T"], w["K"], w["S"], w["B"]
dev = torch.device("cuda:0")
table = synth.make_service_table(T, S, seed=0, degree=32)
net, low, high = build_models(T, S, K, dev, w["n_gcn"])
pipe = ML2PNPipeline(net, low, high, K, precision=a.precision)
svc = DeviceServices.from_table(table, dev)
dbs = [DeviceBatch.from_problems(synth.make_problem_batch(table, B, seed=50 + i, tasks_per_problem=w["n_t"]), dev) for i in range(4)]
pin = lambda t: t.cpu().pin_memory()     # noqa: E731


def host_copy(b):
    return DeviceBatch(pin(b.x), graph.CSR(pin(b.wf_csr.rowptr), pin(b.wf_csr.col), None, b.wf_csr.n), pin(b.seg_ptr),
                       pin(b.local_bounds), pin(b.present), pin(b.global_bounds), b.max_nodes)


hbs = [host_copy(b) for b in dbs]                      # seven pinned tensors per batch: seven host-to-device copies per step
in_bytes = sum(t.numel() * t.element_size() for t in (hbs[0].x, hbs[0].wf_csr.rowptr, hbs[0].wf_csr.col, hbs[0].seg_ptr,
                                                       hbs[0].local_bounds, hbs[0].present, hbs[0].global_bounds))
runner = PipelinedRunner(pipe, svc, dbs[0], slots=2)
pbs = [runner.pack(b) for b in dbs]                     # resident, the slots' layout: one device-to-device copy per step (what bench.py times)
ahs = [runner.pack_host(b) for b in dbs]                # ONE pinned arena per batch in the slots' layout: one host-to-device copy per step
host_idx = [torch.empty(B, T, dtype=torch.int32).pin_memory() for _ in range(2)]
host_R = [torch.empty(B, dtype=torch.float32).pin_memory() for _ in range(2)]
host_out = [torch.empty(B * T + B, dtype=torch.int32).pin_memory() for _ in range(2)]
dev_out = [torch.empty(B * T + B, dtype=torch.int32, device=dev) for _ in range(2)]
out_bytes = host_idx[0].numel() * 4 + host_R[0].numel() * 4


def run(batches, copy_back, steps):
    def two(out, s):
        host_idx[s].copy_(out["idx_high"], non_blocking=True)
        host_R[s].copy_(out["R"], non_blocking=True)

    def one(out, s):                                       # idx_high and R packed on the device, one device-to-host copy
        d = dev_out[s]
        d[: B * T].copy_(out["idx_high"].view(-1), non_blocking=True)
        d[B * T:].copy_(out["R"].view(torch.int32), non_blocking=True)
        host_out[s].copy_(d, non_blocking=True)
    after = {"two": two, "one": one}.get(copy_back)      # runs on the slot's stream right behind its replay (submit's `after`)
    for i in range(8):
        runner.submit(batches[i % 4])
    runner.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        runner.submit(batches[i % 4], after=after)
    runner.synchronize()
    return B * steps / (time.perf_counter() - t0)


resident = max(run(pbs, None, a.steps) for _ in range(2))
seven = max(run(hbs, "two", a.steps) for _ in range(2))
one = max(run(ahs, "one", a.steps) for _ in range(2))
h2d_only = max(run(ahs, None, a.steps) for _ in range(2))          # which direction costs what
d2h_only = max(run(pbs, "one", a.steps) for _ in range(2))
print(f"{a.workload} B={B} {a.precision}: inputs resident in HBM {resident / 1e3:.1f} k problems/s; every batch from pinned host memory and "
      f"results back to pinned host memory: {seven / 1e3:.1f} k problems/s with seven host-to-device copies and two device-to-host "
      f"copies per step, {one / 1e3:.1f} k ({one / resident:.3f} of the resident rate) with ONE copy each way "
      f"(PipelinedRunner.pack_host; {in_bytes / 1024:.0f} KB in, {out_bytes / 1024:.0f} KB out per step of {B} problems); "
      f"host-to-device only {h2d_only / resident:.3f}, device-to-host only {d2h_only / resident:.3f} of the resident rate; "
      f"runner: halves={runner.halves} lockstep={runner.lockstep}")
