#!/usr/bin/env python3
"""ML+2PN inference throughput on MI355X (BASELINE.json metric: service-composition problems/sec).

    python bench.py --gpus N --steps K --warmup W [--workload qws|normal|synth4|synth5] [--batch B]
                    [--precision f32|split|f16] [--scaling weak|strong] [--global-batch G]

A "step" = one pass of the whole hot path over one batch of B synthetic problems that is already
resident in HBM: GNN scores -> per-category top-K feasible candidates -> Low/High pointer-network
encode + greedy decode -> QoS reward (+ at N>1 the single all-gather of the selected indices).
One process per GPU.  Launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the
ranks come from the environment; launched plainly as `python bench.py --gpus N` (N > 1) it starts its own N rank
processes BEFORE anything touches a GPU and exits with their status.  Rank 0 prints ONE JSON line.
  --scaling weak   (default) every rank owns its own B problems per step: global batch = N * B
  --scaling strong the global batch is fixed (--global-batch, default the workload's: 4096 for synth4 = BASELINE
                   configs[3] "batch=4096 sharded over 8") and sharded contiguously: rank r runs G/N problems per step
No data-path collective besides the one all-gather of the selected indices.
The timed region (exactly K steps between barrier + synchronize on both sides, MAX over ranks) is REPEATED until at
least 1 s has been measured; `value` / `ms_per_step` are the median round, `timing` carries min / median / max.
Steps rotate over --batches (default 4) distinct resident batches per rank.

Extra objects on the line:
  roofline     : the dominant kernel by time, priced against the bound that applies to it
  kernels      : every timed kernel of the step (avg ms, algorithmic bytes/flops, fraction of peak), plus — marked
                 "in_step": false — the GCN aggregate on the reference's batching of the service graph (B copies),
                 which the step does not run (cached service embedding) but the north star prices against HBM
  cpu_baseline : the CPU oracle (oracle/, a torch-CPU port of the reference algorithm) timed on this
                 box's host cores over a bounded sample of the same workload (rank 0, N=1 only)
  other_precision : a second measurement of the same workload in the OTHER arithmetic of the recurrent W_hh.h products —
                 the fp32 matrix cores when `value` is the exact split (the default), the exact split when `value` is f32 —
                 with its own kernel table and the agreement of the two results; reported beside `value`, never as `value`
Defaults: --gpus 1 --steps 100 --warmup 10 --workload qws (a few seconds of GPU time + ~10 s of CPU baseline).
"""
import argparse
import contextlib
import gc
import json
import os
import sys
import time

# Hardware queues (round 6, profiles/LOG_r06.md section 9): HIP deals a process's streams onto GPU_MAX_HW_QUEUES (default 4) in-order
# hardware queues round robin, and in a rank of a process group the all-gather's "wait for slot A's step" sat in front of slot B's
# launches (head-of-line blocking: -6 % with a real RCCL collective every 8 steps).  Nothing is set here: PipelinedRunner gives its
# slots' streams a priority of their own wherever a collective stream exists (pipeline.py), which keeps them out of the process
# group's queue with the default 4 queues — asking for 8 queues does the same (and this file did, until the library learnt to),
# but 8 queues AND the priority is far worse than either (-22 %), and 8 queues without a process group cost the PCIe-inclusive
# leg a third of its rate.

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X peaks (/opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0          # HBM3E spec; 6.29 TB/s is the measured copy ceiling
PEAK_F32_TFLOPS = 157.3        # fp32 matrix (= fp32 vector) dense peak
PEAK_F16_TFLOPS = 2500.0       # fp16 / bf16 MFMA dense peak (never the 2:1-sparsity figure)

WORKLOADS = {
    # name: T, K, S, per-GPU batch, task nodes per problem, GCN layers   
    "qws": dict(T=47, K=5, S=2507, B=256, n_t=10, n_gcn=2,
                desc="QWS shape: T=47 K=5 L=235 S=2507(assumed) H=256, batch=256 (BASELINE configs[1])"),
    "normal": dict(T=50, K=10, S=5000, B=1024, n_t=10, n_gcn=4,
                   desc="Normal shape: T=50 K=10 L=500 S=5000(assumed) H=256, batch=1024 (configs[2])"),
    "synth4": dict(T=1000, K=5, S=5000, B=512, n_t=1000, n_gcn=2, G=4096,
                   desc="synthetic 1000-task/5000-candidate, batch=512 per GPU (configs[3] = 4096 over 8)"),
    "synth5": dict(T=2000, K=10, S=20000, B=256, n_t=2000, n_gcn=2,
                   desc="synthetic 2000-task/20000-candidate, batch=256 per GPU (configs[4]; its 'fp16 encoder MFMA path' "
                        "is --precision f16)"),
}


def build_models(T, S, K, dev, n_gcn=2, hidden_pn=256, seed=0):
    """Random-init weights of the reference architecture (there are no checkpoints offline):
    Net per environment.ini:[QWS-ML]/[Normal-ML], two CombinatorialRL per [*-PNHigh]."""
    from gnnpn_sc_amd.modelML import Net
    from gnnpn_sc_amd.modelPN import CombinatorialRL, reward
    torch.manual_seed(seed)
    net = Net(128, S, 20, 2, n_gcn, vocab=max(100, T + 1))
    low = CombinatorialRL(0, hidden_pn, T * K, 0, 10, 1, reward, "Dot", K, T, level="Low")
    high = CombinatorialRL(0, hidden_pn, T * K, 0, 10, 1, reward, "Dot", K, T, level="High")
    return net.to(dev).eval(), low.to(dev).eval(), high.to(dev).eval()


def algorithmic_cost(name, w, B, precision="f32"):
    """Algorithmic bytes / flops of ONE launch of each timed kernel."""
    T, K, H = w["T"], w["K"], 256
    L = T * K
    if name == "lstm_encode":       # both nets in one launch: recurrent matmul flops, priced against the matrix peak of the
        # operand type the products run in: fp32 MFMA | fp16 MFMA (f16: one product; split: three fp16 products per term)
        flops = 2 * B * L * 2 * H * 4 * H
        if precision == "f16":
            return dict(bound="mfma", work=flops, unit="TFLOP/s", peak=PEAK_F16_TFLOPS)
        if precision == "split":   # exact split: six fp16 products per fp32 term (coop_common.h) — the EXECUTED flops on the f16 cores
            return dict(bound="mfma", work=6 * flops, unit="TFLOP/s", peak=PEAK_F16_TFLOPS, fp32_equivalent_work=flops)
        return dict(bound="mfma", work=flops, unit="TFLOP/s", peak=PEAK_F32_TFLOPS)
    if name == "pointer_decode":    # BOTH nets in one launch; per net and problem exactly SURVEY.md section 8d's "PN decode,
        # one problem, one net": every enc_out row once (L*H*4) + per-step state T*(2*H*4) + outputs T*(8 + K*4)
        byt = 2 * B * (L * H * 4 + T * (2 * H * 4) + T * (8 + K * 4))
        return dict(bound="hbm", work=byt, unit="GB/s", peak=PEAK_HBM_GBS)
    if name == "request_branch":    # fused GIN branch: reads the node rows + CSR once, writes [B,128]; weights are L2-resident
        return dict(bound="hbm", work=w["n_nodes"] * 7 * 4 + w["n_edges"] * 4 + B * 128 * 4, unit="GB/s", peak=PEAK_HBM_GBS)
    if name == "pregates_gemm":     # [B*L,256] x [256,1024] per net
        return dict(bound="mfma", work=2 * B * L * H * 4 * H, unit="TFLOP/s", peak=PEAK_F32_TFLOPS)
    if name == "csr_aggregate_gcn":  # 2*S*C*4 + E*(4+4) + (S+1)*4, C=256
        S, E = w["S"], w["E"]
        return dict(bound="hbm", work=2 * S * 256 * 4 + E * 8 + (S + 1) * 4, unit="GB/s", peak=PEAK_HBM_GBS)
    raise KeyError(name)


class KernelTimers:
    """HIP-event pairs around chosen C-ABI calls, recorded on the stream the kernels run on
    (torch's current stream, which is the stream handed to the C ABI)."""

    def __init__(self):
        self.events = {}
        self.enabled = False

    def wrap(self, module, fn_name, label, select=None):
        orig = getattr(module, fn_name)

        def timed(*a, **k):
            if not self.enabled or (select is not None and not select(*a, **k)):
                return orig(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = orig(*a, **k)
            e1.record()
            self.events.setdefault(label, []).append((e0, e1))
            return out
        setattr(module, fn_name, timed)

    def summary(self):
        if os.environ.get("GNNPN_BENCH_DEBUG"):
            for k, v in self.events.items():
                print(k, [round(a.elapsed_time(b), 3) for a, b in v], file=sys.stderr)
        return {k: (sum(a.elapsed_time(b) for a, b in v) / len(v), len(v)) for k, v in self.events.items()}


def _oracle_pass(ctx, lo, hi):
    """The CPU oracle chain (oracle/: torch-CPU fp32 restatement of the reference algorithm) over problems [lo, hi) of the
    batch: GNN forward, stable ranking, candidate reduction, Low+High greedy decode with full-L attention, reward."""
    import numpy as np
    from oracle import ml as oml       # checker / baseline use only
    from oracle import pn as opn
    import gnnpn_sc_amd.synth as synth
    from gnnpn_sc_amd.loadData import reduce_from_ranking
    w, table, pb = ctx["w"], ctx["table"], ctx["pb"]
    T, K = w["T"], w["K"]
    nodes_per = pb.x.shape[0] // pb.n_problems
    n0, n1 = lo * nodes_per, hi * nodes_per
    em = (pb.edge_index[0] >= n0) & (pb.edge_index[0] < n1)
    sub = synth.ProblemBatch(pb.x[n0:n1], pb.edge_index[:, em] - n0, pb.batch[n0:n1] - lo, pb.local_bounds[lo:hi],
                             pb.present[lo:hi], pb.global_bounds[lo:hi])
    data = oml.make_data(torch.from_numpy(sub.x), torch.from_numpy(sub.edge_index), torch.from_numpy(sub.batch),
                         torch.from_numpy(table.x_service), torch.from_numpy(table.edge_index),
                         torch.from_numpy(table.edge_attr))
    scores = oml.net_forward(ctx["sd_ml"], data, 2, w["n_gcn"])
    rank = oml.rank_services(scores).numpy()
    cat_of = np.repeat(np.arange(T), np.diff(table.cat_ptr))
    rows = [reduce_from_ranking(rank[b], sub.local_bounds[b], sub.present[b], sub.global_bounds[b], cat_of,
                                table.qos, K) for b in range(hi - lo)]
    x = torch.tensor(rows, dtype=torch.float32)[:, :, 1:]
    return opn.two_level_greedy(ctx["sd_low"], ctx["sd_high"], x, T, K)


def _cpu_worker(ctx, start, B, threads, seconds, barrier, q):
    """One process of the process-parallel CPU baseline: passes of up to 64 problems over the batch, beginning at ITS offset
    `start` (the same pass size as the single-process run reaches: the oracle recomputes the service branch per pass)."""
    torch.set_num_threads(threads)
    chunk = min(B, 64)
    _oracle_pass(ctx, start, min(B, start + 2))          # warm-up: thread pool, allocator, first-touch
    barrier.wait()
    t0, done, pos = time.time(), 0, start
    while True:
        e = min(B, pos + chunk)
        _oracle_pass(ctx, pos, e)
        done += e - pos
        pos = 0 if e >= B else e
        if time.time() - t0 >= seconds:
            break
    q.put((t0, time.time(), done))


def usable_cpus():
    """CPUs this process may use: its affinity mask, capped by the cgroup's CPU quota where one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(w, table, pb, sd_ml, sd_low, sd_high, budget_s=10.0):
    """The CPU oracle chain on a bounded sample of the same batch, on this box's host cores — called BEFORE the process
    touches a GPU (it starts worker processes).  Problems are independent, so the honest figure for "the same box's host
    cores" is the best of: one process with every usable core as threads, and P processes x t threads over disjoint
    problems (the per-step LSTM matmuls are tiny and stop scaling with threads long before a many-core host is full).
    `value` = the best configuration measured; the 16-thread and 1-thread single-process figures of rounds 1-2 stay beside it."""
    import multiprocessing as mp
    B = pb.n_problems
    ctx = {"w": {k: w[k] for k in ("T", "K", "n_gcn")}, "table": table, "pb": pb, "sd_ml": sd_ml, "sd_low": sd_low, "sd_high": sd_high}
    ncpu = usable_cpus()

    def single(threads, seconds):
        torch.set_num_threads(threads)
        n = min(B, 8)
        t0 = time.perf_counter()
        _oracle_pass(ctx, 0, n)
        t_cal = time.perf_counter() - t0
        n = int(max(n, min(B, n * seconds / max(t_cal, 1e-3) / 2)))
        passes, t0 = 0, time.perf_counter()
        while True:
            _oracle_pass(ctx, 0, n)
            passes += 1
            el = time.perf_counter() - t0
            if el >= seconds or passes >= 16:
                break
        return {"processes": 1, "threads_per_process": threads, "cores": threads, "value": round(n * passes / el, 2),
                "sample": f"{passes} pass(es) over the first {n} problems, {el:.1f} s"}

    def parallel(P, t, seconds):
        mpc = mp.get_context("spawn")
        barrier, q = mpc.Barrier(P), mpc.Queue()
        per = max(1, B // P)
        procs = [mpc.Process(target=_cpu_worker, args=(ctx, (i * per) % B, B, t, seconds, barrier, q)) for i in range(P)]
        for pr in procs:
            pr.start()
        res = [q.get(timeout=600) for _ in procs]
        for pr in procs:
            pr.join(timeout=60)
        wall = max(r[1] for r in res) - min(r[0] for r in res)
        done = sum(r[2] for r in res)
        return {"processes": P, "threads_per_process": t, "cores": P * t, "value": round(done / wall, 2),
                "sample": f"{done} problems in {wall:.1f} s: every process runs passes of <= 64 problems over the batch from its own offset"}

    configs = []
    t16 = min(ncpu, 16)
    configs.append(single(t16, budget_s * 0.4))
    print(f"[cpu_baseline] host has {os.cpu_count()} CPUs, {ncpu} usable; 1 x {t16} threads: {configs[-1]['value']} problems/s",
          file=sys.stderr, flush=True)
    if ncpu > 16:
        configs.append(single(ncpu, budget_s * 0.3))
        print(f"[cpu_baseline] 1 x {ncpu} threads: {configs[-1]['value']} problems/s", file=sys.stderr, flush=True)
    for t in (8, 2):
        P = min(ncpu // t, B, 128)
        if P > 1:
            try:
                configs.append(parallel(P, t, budget_s * 0.4))
                print(f"[cpu_baseline] {P} x {t} threads: {configs[-1]['value']} problems/s", file=sys.stderr, flush=True)
            except Exception as e:      # noqa: BLE001  (a box that refuses that many processes: keep what was measured)
                print(f"[cpu_baseline] {P} x {t} threads failed: {e!r}", file=sys.stderr, flush=True)
    one = single(1, budget_s * 0.2) if ncpu > 1 else None
    best = max(configs, key=lambda c: c["value"])
    torch.set_num_threads(min(ncpu, 16))
    return {"value": best["value"], "unit": "problems/s", "cores": best["cores"], "kind": "port",
            "processes": best["processes"], "threads_per_process": best["threads_per_process"],
            "host_cpus": os.cpu_count(), "usable_cpus": ncpu, "configurations": configs, "single_thread": one,
            "sample": f"best of {len(configs)} configurations ({best['processes']} process(es) x {best['threads_per_process']} threads: "
                      f"{best['sample']}) of the same batch through oracle/ (torch-CPU fp32 port of the reference algorithm: GNN "
                      f"forward, stable ranking, candidate reduction, Low+High greedy decode with full-L attention, reward); "
                      f"torch {torch.__version__}"}


REAL_STDOUT = 1


def batched_aggregate_roofline(table, B, dev, reps=10):
    """The GCN aggregate on the reference's own batching of the service graph (B block-diagonal copies of the table,
    trainML.py:109-114, modelML.py:145-156): one layer, 256 channels, through ops.csr_aggregate (the tiled kernel — one source
    tile's 16-channel slice in LDS at a time — where the graph's plan is valid, as it is for the reference's edge order).  NOT part of the step (the step uses the cached
    service embedding): reported in ``kernels`` with "in_step": false so that the north star's aggregate roofline has a
    number in this line.  HIP events on the stream the launches go to (torch's current stream), best of 3 rounds."""
    from gnnpn_sc_amd import graph, ops
    import gnnpn_sc_amd.synth as synth
    S = table.n_services
    copies = max(1, min(B, 256, 700_000 // S))
    # the table's graph with its edge list in the order the reference's scan emits it (src/loadData.py:56-65: rows sorted by
    # source) — the order the reference's own service graphs have and the tiled aggregate needs
    ei, ea = synth.scan_order(table.edge_index, table.edge_attr)
    csr = graph.gcn_csr(torch.from_numpy(ei), torch.from_numpy(ea), S)
    nnz = csr.col.numel()
    rp = torch.cat([csr.rowptr[:-1].long() + c * nnz for c in range(copies)] + [torch.tensor([copies * nnz])]).int().to(dev)
    col = torch.cat([csr.col.long() + c * S for c in range(copies)]).int().to(dev)
    norm = ops.gcn_norm(rp, col, csr.w.repeat(copies).to(dev))
    N, C = copies * S, 256
    x = torch.randn(N, C, device=dev)
    bias = torch.zeros(C, device=dev)
    run = lambda: ops.csr_aggregate(rp, col, norm, x, bias=bias, act=ops.ACT_RELU, block_rows=S)   # noqa: E731
    best = float("inf")
    for rnd in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            best = min(best, e0.elapsed_time(e1) / reps)
    work = 2 * N * C * 4 + copies * nnz * 8 + (N + 1) * 4             # SURVEY section 8d per graph, times the copies
    plan = ops._tile_plans.get((id(rp), id(col), id(norm), S))
    form = ("tiled" if plan is not None and plan[2].valid else
            "lds-staged" if S <= ops.LDS_SLICE16_ROWS_MAX and copies * (C // 16) >= ops.LDS_MIN_WORKGROUPS else "l2-gather")
    return {"kernel": "csr_aggregate_batched", "in_step": False, "form": form, "edge_order": "reference scan order (src/loadData.py:56-65)",
            "copies": copies, "rows": N, "edges": copies * nnz,
            "launches_per_step": 0, "avg_ms": round(best, 4), "bound": "hbm", "achieved": round(work / best / 1e6, 3),
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(work / best / 1e6 / PEAK_HBM_GBS, 5)}


def pcie_inclusive_rate(runner, batches, B, T, steps, dev):
    """The step rate when every batch STARTS in pinned host memory and the selected indices and scores go back to pinned host
    memory (what a caller of the reference's Python API holds: src/models/trainPNHigh.py:134-136 `inputs.cuda()`, :140-141
    `.cpu()`): one host-to-device copy of the batch's arena (PipelinedRunner.pack_host) and one device-to-host copy of the
    packed results per step, on the slot's stream.  Reported beside `value`, never as `value`."""
    hosts = [runner.pack_host(b) for b in batches]
    host_out = [torch.empty(B * T + B, dtype=torch.int32).pin_memory() for _ in range(runner.n_slots)]
    dev_out = [torch.empty(B * T + B, dtype=torch.int32, device=dev) for _ in range(runner.n_slots)]

    def results_home(out, s):             # on the slot's stream, right behind its replay (PipelinedRunner.submit's `after`)
        d = dev_out[s]
        d[: B * T].copy_(out["idx_high"].reshape(-1), non_blocking=True)
        d[B * T:].copy_(out["R"].view(torch.int32), non_blocking=True)
        host_out[s].copy_(d, non_blocking=True)

    def run(n):
        for i in range(n):
            runner.submit(hosts[i % len(hosts)], after=results_home)
        runner.synchronize(check=False)
    run(8)
    best = 0.0
    gc.disable()
    for _ in range(2):                # the first pass still pays the first use of the pinned arenas
        t0 = time.perf_counter()
        run(steps)
        best = max(best, B * steps / (time.perf_counter() - t0))
    gc.enable()
    return best


def gpu_identity(local_rank):
    """Which physical card a rank runs on — the card's unique id and PCI location from the KFD topology in sysfs (the GPU nodes this
    process may open, in node order = HIP's device order; no exec, no GPU call): hand-off time-outs have been card-dependent
    (profiles/LOG_r01_r04.md section 13.3), so every rank says where it ran (`per_rank` of the bench line)."""
    import glob
    try:
        gpus = []
        for d in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda p: int(os.path.basename(p))):
            props = {}
            try:
                with open(os.path.join(d, "properties")) as f:
                    for ln in f:
                        k, _, v = ln.partition(" ")
                        props[k] = v.strip()
            except OSError:                     # a node of another tenant's card: not readable from this container
                continue
            if int(props.get("simd_count", "0")) > 0:
                loc = int(props.get("location_id", "0"))
                gpus.append({"unique_id": hex(int(props["unique_id"])) if props.get("unique_id", "0") != "0" else None,
                             "pci": f"{int(props.get('domain', '0')):04x}:{loc >> 8:02x}:{(loc >> 3) & 31:02x}.{loc & 7}"})
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
            vis = os.environ.get(var)
            if vis and all(v.strip().isdigit() for v in vis.split(",")) and all(int(v) < len(gpus) for v in vis.split(",")):
                gpus = [gpus[int(v)] for v in vis.split(",")]
        return gpus[local_rank] if local_rank < len(gpus) else None
    except (OSError, ValueError, KeyError):
        return None


def bind_to_gpu_numa_node(local_rank):
    """Pin this rank's process to the CPUs of the NUMA node its GPU hangs off (eight ranks replaying two HIP graphs every
    0.4-0.8 ms are launch-rate-sensitive: a rank whose host threads sit on the far socket pays for every doorbell).  Reads
    sysfs only — it runs BEFORE anything touches the GPU, and it neither execs nor re-launches.  The AMD GPUs are taken in
    PCI-address order (the order the HIP runtime enumerates them in), filtered through HIP_/ROCR_VISIBLE_DEVICES when those
    are plain index lists.  Returns what it did, for the bench line."""
    import glob
    import re
    try:
        cards = []
        for d in glob.glob("/sys/class/drm/card[0-9]*"):
            if not re.fullmatch(r"card\d+", os.path.basename(d)):
                continue
            dev = os.path.join(d, "device")
            with open(os.path.join(dev, "vendor")) as f:
                if f.read().strip() != "0x1002":
                    continue
            with open(os.path.join(dev, "numa_node")) as f:
                node = int(f.read().strip())
            cards.append((os.path.basename(os.path.realpath(dev)), node))
        cards.sort()
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
            vis = os.environ.get(var)
            if vis and all(v.strip().isdigit() for v in vis.split(",")):
                cards = [cards[int(v)] for v in vis.split(",") if int(v) < len(cards)]
        ident = gpu_identity(local_rank)      # the KFD topology lists the cards THIS process may open (a shared host shows every card
        pci = node = None                     # under /sys/class/drm): its PCI address wins where it is known
        if ident is not None:
            try:
                with open(f"/sys/bus/pci/devices/{ident['pci']}/numa_node") as f:
                    pci, node = ident["pci"], int(f.read().strip())
            except (OSError, ValueError):
                pci = node = None
        if pci is None:
            if local_rank >= len(cards):
                return {"bound": False, "why": f"{len(cards)} AMD GPU(s) in sysfs, local rank {local_rank}"}
            pci, node = cards[local_rank]
        if node < 0:
            return {"bound": False, "gpu_pci": pci, "why": "the GPU reports no NUMA node"}
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            cpus = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return {"bound": False, "gpu_pci": pci, "numa_node": node, "why": "none of the node's CPUs is in this process's mask"}
        os.sched_setaffinity(0, cpus)
        return {"bound": True, "gpu_pci": pci, "numa_node": node, "cpus": len(cpus)}
    except (OSError, ValueError, IndexError) as e:
        return {"bound": False, "why": repr(e)}


_share_lock_file = None


@contextlib.contextmanager
def gpu_turn(share):
    """Ranks that share ONE GPU (GNNPN_BENCH_SHARE_GPU=1: a rehearsal of the launch path, never a measurement) take turns on it:
    the cooperative kernels place their workgroups per process (seat table, staffing count), and launches of four processes
    on one card are a mix nobody designed for — one such run ended in a hand-off time-out.  A file lock around every stretch
    of GPU work (never around a collective: the peers could not reach it) keeps one rank's launches on the card at a time."""
    global _share_lock_file
    if not share:
        yield
        return
    import fcntl
    if _share_lock_file is None:
        _share_lock_file = open(f"/tmp/gnnpn_bench_share_{os.environ.get('MASTER_PORT', '0')}.lock", "w")
    fcntl.flock(_share_lock_file, fcntl.LOCK_EX)
    try:
        yield
    finally:
        torch.cuda.synchronize()
        fcntl.flock(_share_lock_file, fcntl.LOCK_UN)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU) and relay their status.
    Runs BEFORE this process touches a GPU (torch.cuda.device_count() does not initialise one on this image) and never
    replaces a process image: the children are ordinary subprocesses, the parent only waits."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    share = os.environ.get("GNNPN_BENCH_SHARE_GPU") == "1"    # launch-path check on a box with fewer GPUs: ranks share
    if have < n and not share:                                 # GPU (rank % count) and the collective runs over gloo
        raise SystemExit(f"--gpus {n}: this node has {have} GPU(s)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    raise SystemExit(rc)


def rank_batches(synth, table, w, rank, world, scaling, n_batches):
    """The resident synthetic batches of one rank (host side).  weak: B problems per rank and batch, seeds differ by
    rank.  strong: batch j is ONE global set of G problems made of 8 (or `world`) seeded chunks, of which this rank
    owns a contiguous run — the same global set at every N, so N=1 and N=8 work on identical problems."""
    out = []
    if scaling == "weak":
        for j in range(n_batches):
            out.append(synth.make_problem_batch(table, w["B"], seed=1 + rank + 97 * j, tasks_per_problem=w["n_t"]))
        return out
    import numpy as np
    n_chunks = 8 if 8 % world == 0 else world
    G = w["G"]
    if G % n_chunks:
        raise SystemExit(f"--global-batch {G} must be a multiple of {n_chunks}")
    per = n_chunks // world
    for j in range(n_batches):
        parts = [synth.make_problem_batch(table, G // n_chunks, seed=1 + 97 * j + 1009 * c, tasks_per_problem=w["n_t"])
                 for c in range(rank * per, (rank + 1) * per)]
        nodes = np.cumsum([0] + [p.x.shape[0] for p in parts])
        probs = np.cumsum([0] + [p.n_problems for p in parts])
        out.append(synth.ProblemBatch(
            np.concatenate([p.x for p in parts]), np.concatenate([p.edge_index + nodes[i] for i, p in enumerate(parts)], 1),
            np.concatenate([p.batch + probs[i] for i, p in enumerate(parts)]),
            np.concatenate([p.local_bounds for p in parts]), np.concatenate([p.present for p in parts]),
            np.concatenate([p.global_bounds for p in parts])))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="qws", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="problems per GPU and step (default: the workload's; weak scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: problems per step over ALL GPUs "
                    "(default: the workload's G, else its per-GPU batch)")
    ap.add_argument("--batches", type=int, default=4, help="distinct resident batches the steps rotate over")
    ap.add_argument("--min-time", type=float, default=1.0, help="repeat the K-step timed region until this many seconds")
    ap.add_argument("--settle", type=int, default=24, help="at most this many UNTIMED rounds of K steps behind the warm-up, until three in a row "
                    "agree within 2 %% (0: none — the first timed round follows the W warm-up steps directly); they are listed in timing.settling_round_us")
    ap.add_argument("--write-through", type=int, default=-1, help="hand-off form of the cooperative kernels: 1 = agent-scope write-through "
                    "granule stores (placement independent), 0 = stores that stay in the XCD's L2 (same-XCD groups), -1 = the library's default")
    ap.add_argument("--precision", default="split", choices=["f32", "f16", "split"],
                    help="arithmetic of the recurrent W_hh.h products.  split (default): fp32 operands decomposed EXACTLY into "
                         "three fp16 pieces, every cross term >= 2^-24 kept, fp32 accumulate — fp32 in, fp32 out, no operand "
                         "bit dropped (DESIGN.md section 5; profiles/LOG_r01_r04.md section 12); f32: the fp32 matrix cores; f16: opt-in fp16-operand encoder "
                         "(BASELINE configs[4]), NOT parity-exact — the line then carries the agreement with the f32 path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-split-line", "--no-other-precision", dest="no_split_line", action="store_true",
                    help="skip the second measurement (same workload in the other arithmetic: f32 <-> split)")
    ap.add_argument("--graph", type=int, default=1,
                    help="1 (default): replay each step as one captured HIP graph, per-kernel HIP events in a "
                         "separate eager pass; 0: eager launches with the HIP events inside the timed region "
                         "(the events themselves cost ~25 %% throughput)")
    ap.add_argument("--gather-every", type=int, default=0,
                    help="N > 1: steps of a slot whose selected indices travel in ONE all-gather (0 = 8 at the QWS shape, else 1)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="independent steps (batches) in flight on separate HIP streams (needs --graph 1)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                                 # never returns
    # stdout carries ONE line, the JSON: the libraries initialised below write to the C stdout (RCCL's version banner, gloo's
    # connection notes), so fd 1 is pointed at stderr for the run and the real one is kept for the line
    global REAL_STDOUT
    sys.stdout.flush()
    REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    share = os.environ.get("GNNPN_BENCH_SHARE_GPU") == "1"
    affinity = bind_to_gpu_numa_node(0 if share else local_rank) if world > 1 or os.environ.get("GNNPN_BENCH_BIND") == "1" else None
    if not share and torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):   # device_count() does not initialise a GPU
        raise SystemExit(f"rank {rank}: {torch.cuda.device_count()} GPU(s) visible, {os.environ.get('LOCAL_WORLD_SIZE', world)} ranks on this node")
    import gnnpn_sc_amd.synth as synth
    w = dict(WORKLOADS[args.workload])
    if args.batch:
        w["B"] = args.batch
    if args.scaling == "strong":
        w["G"] = args.global_batch or w.get("G", w["B"])
        if w["G"] % world:
            raise SystemExit(f"--global-batch {w['G']} is not divisible by {world} GPUs")
        w["B"] = w["G"] // world
    T, K, S, B = w["T"], w["K"], w["S"], w["B"]
    table = synth.make_service_table(T, S, seed=0, degree=32)
    w["E"] = int(table.edge_index.shape[1]) + S            # + self loops
    host_batches = rank_batches(synth, table, w, rank, world, args.scaling, max(1, args.batches))
    pb = host_batches[0]
    w["n_nodes"], w["n_edges"] = int(pb.x.shape[0]), int(pb.edge_index.shape[1])

    # The CPU baseline runs FIRST, before this process touches a GPU: it starts worker processes (problems are independent:
    # P processes x t threads is the honest use of the host's cores), which a GPU-initialised process must not do.
    # Same seeded weights as the GPU run below (build_models seeds torch before constructing them).
    cpu_line = None
    if world == 1 and not args.no_cpu_baseline:
        net_c, low_c, high_c = build_models(T, S, K, torch.device("cpu"), w["n_gcn"])
        cpu_line = cpu_baseline(w, table, pb, *({k: v.detach().clone() for k, v in m.state_dict().items()} for m in (net_c, low_c, high_c)))
        del net_c, low_c, high_c

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ML+2PN hot path has no CPU implementation")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count() if share else local_rank)
    torch.cuda.set_device(dev)

    from gnnpn_sc_amd import dist as gdist
    from gnnpn_sc_amd import ops
    from gnnpn_sc_amd.pipeline import DeviceBatch, DeviceServices, ML2PNPipeline, PipelinedRunner

    force_dist = os.environ.get("GNNPN_FORCE_DIST") == "1"     # exercise the RCCL path with one rank
    use_dist = world > 1 or force_dist
    first_collective_ms = None
    if use_dist:
        gdist.init_process_group("gloo" if share else "nccl", None if share else dev)
        # the process group's FIRST collective (communicator set-up, ring discovery over xGMI) and a second one of the step's
        # shape, timed on the host: a slow 8-GPU line whose cause is the collective, not the kernels, shows here (`per_rank`)
        import time as _t
        probe = torch.zeros((B, T), dtype=torch.int32, device=dev)
        t0 = _t.perf_counter()
        g0, w0 = gdist.all_gather_indices_async(probe, None)
        if w0 is not None:
            w0.wait()
        torch.cuda.synchronize(dev)
        t1 = _t.perf_counter()
        g0, w0 = gdist.all_gather_indices_async(probe, g0)
        if w0 is not None:
            w0.wait()
        torch.cuda.synchronize(dev)
        first_collective_ms = {"first": round((t1 - t0) * 1e3, 3), "second": round((_t.perf_counter() - t1) * 1e3, 3)}
        del probe, g0

    net, low, high = build_models(T, S, K, dev, w["n_gcn"])
    pipe = ML2PNPipeline(net, low, high, K, precision=args.precision)
    svc = DeviceServices.from_table(table, dev)
    batches = [DeviceBatch.from_problems(b, dev) for b in host_batches]
    batch = batches[0]
    del host_batches[1:]

    timers = KernelTimers()
    if not args.no_kernel_timers:
        # the mirrors reach the kernels through the C++ operators: the two recurrent launches and the GCN aggregate through
        # custom_ops' callers, the rest as torch.ops.gnnpn.* (an attribute of the namespace object: wrapped the same way)
        from gnnpn_sc_amd import custom_ops
        timers.wrap(custom_ops, "lstm_encode", "lstm_encode")
        timers.wrap(custom_ops, "pointer_decode", "pointer_decode")
        timers.wrap(torch.ops.gnnpn, "linear", "pregates_gemm", select=lambda a, wt, *r, **k: tuple(wt.shape) == (1024, 256))
        timers.wrap(custom_ops, "csr_aggregate", "csr_aggregate_gcn", select=lambda rp, c, wv, *r, **k: wv is not None)
        timers.wrap(torch.ops.gnnpn, "request_branch", "request_branch")

    # Steps are independent batches.  Default: pipeline.PipelinedRunner — `inflight` slots, each one
    # captured HIP graph of the whole pass with its own stream, static inputs/outputs and hand-off workspaces; step i
    # copies resident batch i % n_batches into slot i % inflight's static inputs (device-to-device, on the slot's
    # stream, inside the timed region) and replays the slot's graph, so the latency-bound recurrent kernels of one
    # step overlap the next step's kernels.  Events cannot be read back from a replayed graph, and event pairs between
    # the kernels of an eager timed region cost ~25 % throughput, so the per-kernel durations come from a
    # separate eager pass on one stream right after the timed region; --graph 0 (eager, one stream)
    # records them inside the timed region instead.
    n_slots = max(1, args.inflight) if args.graph else 1
    if share:       # ranks sharing ONE GPU (launch-path rehearsal): a CU holds two cooperative workgroups in all, so every rank
        n_slots = 1  # runs one launch at a time — four launches of two processes would wait on each other past the spin bound
    with gpu_turn(share):
        runner = PipelinedRunner(pipe, svc, batch, slots=n_slots, halves=False if share else None, auto_degrade=False,   # the degraded form is this file's own (below)
                                 write_through=None if args.write_through < 0 else bool(args.write_through)) if args.graph else None
    if runner is not None:
        batches = [runner.pack(b) for b in batches]             # the slots' layout: one device-to-device copy per step instead of seven
        batch = batches[0]
    last = {}                                                   # slot -> index of the batch its outputs belong to
    gathers = {}                                                # (runner, slot) -> (gathered tensor, pending work)
    # Bucketed collective: the selected indices of `bucket` consecutive steps of a slot cross xGMI as ONE all-gather
    # (48 KB per rank and step at QWS: a collective per 0.55 ms step is all launch overhead — host and device —, 8 steps
    # per collective carry the same bytes in an eighth of the launches).  Two staging buffers per slot: one fills while
    # the other's gather is in flight.  Every gather, the partial last bucket included, lands inside the timed region.
    bucket = args.gather_every if args.gather_every > 0 else (8 if B * T * K <= 256 * 235 * 2 else 1)
    stages = {}                                                 # (runner, slot) -> {"buf": [2 x [bucket,B,T]], "cur", "fill", "out": [2], "work": [2]}

    cur = {"runner": runner, "degraded": None}                  # the headline runner may be replaced by its degraded form (below)

    def step(i, runner=None):
        """One step.  At N > 1 the single collective of the path — the all-gather of the selected indices — is issued
        asynchronously behind the slot's decode: the process group's own stream carries it over xGMI while the slot's
        stream goes straight on to its next step; the slot only waits for it (stream-side) right before the replay that
        would overwrite the gathered buffer, two steps later."""
        j = i % len(batches)
        serial = cur["degraded"] is not None and (runner is None or runner is cur["runner"])
        runner = cur["runner"] if runner is None else runner
        if runner is not None:
            s = runner.count % runner.n_slots
            key = (id(runner), s)
            if use_dist and gathers.get(key, (None, None))[1] is not None:
                with torch.cuda.stream(runner.stream(s)):
                    gathers[key][1].wait()                      # the previous gather out of this slot's index buffer is done
            with gpu_turn(share):
                out, s = runner.submit(batches[j])
            stream = runner.stream(s)
            last[key] = j
        else:
            with gpu_turn(share):
                out = pipe.run(svc, batches[j])
            stream, key = torch.cuda.current_stream(), (0, 0)
        step.last = out
        if use_dist and bucket > 1:
            st = stages.get(key)
            if st is None:
                idx = out["idx_high"]
                st = stages[key] = {"buf": [torch.empty((bucket,) + tuple(idx.shape), dtype=idx.dtype, device=idx.device) for _ in range(2)],
                                    "cur": 0, "fill": 0, "out": [None, None], "work": [None, None], "stream": stream}
            with torch.cuda.stream(stream):
                st["buf"][st["cur"]][st["fill"]].copy_(out["idx_high"], non_blocking=True)
                st["fill"] += 1
                if st["fill"] == bucket:
                    flush_bucket(st)
                    if serial:                                   # degraded rank: the slot's next launches wait for the collective
                        for k in (0, 1):
                            if st["work"][k] is not None:
                                st["work"][k].wait()
                                st["work"][k] = None
            return out["idx_high"], out["R"]
        if use_dist:
            with torch.cuda.stream(stream):
                gathers[key] = gdist.all_gather_indices_async(out["idx_high"], gathers.get(key, (None, None))[0])
                if serial and gathers[key][1] is not None:
                    gathers[key][1].wait()
                    gathers[key] = (gathers[key][0], None)
            return gathers[key][0], out["R"]
        return out["idx_high"], out["R"]

    def seat_totals(r):
        """Declined / off-canonical seats summed over every cooperative launch since the status blocks were last cleared, with the
        encoder workgroup-tiles booked over the same launches (8 x nets x ceil(problems / 16) per launch) as the denominator."""
        tot = {"declined": 0, "off_canonical": 0, "encoder_tiles_booked": 0}
        for wsp in r.workspaces:
            for k in ("declined", "off_canonical"):
                tot[k] += (wsp.last_seats or {}).get(k, 0)
            tot["encoder_tiles_booked"] += ((wsp.last_progress or {}).get("encoder") or {}).get("expected", 0)
        launches = tot["encoder_tiles_booked"] / max(1, ops.Workspaces.coop_units(2, B if r.n_slots > 1 or not getattr(r, "halves", False) else B // 2))
        tot["declined_per_encoder_launch"] = round(tot["declined"] / launches, 4) if launches else None
        return tot

    def flush_bucket(st):
        """Gather the filled part of the current staging buffer (called on the slot's stream) and switch to the other one,
        whose previous gather must have finished before it is written again."""
        c, n = st["cur"], st["fill"]
        if n:
            src = st["buf"][c][:n].reshape(n * st["buf"][c].shape[1], -1)
            if n == bucket:
                st["out"][c], st["work"][c] = gdist.all_gather_indices_async(src, st["out"][c])
            else:                                               # the partial last bucket of a round: its own (smaller) destination
                st["last"], st["work"][c] = gdist.all_gather_indices_async(src, None)
            st["gathered_rows"] = (st["out"][c] if n == bucket else st["last"]).shape[0]
        st["cur"], st["fill"] = 1 - c, 0
        if st["work"][1 - c] is not None:
            st["work"][1 - c].wait()                            # stream-side: the other buffer's gather is done before this slot refills it
            st["work"][1 - c] = None

    def finish_gathers():
        for g, work in gathers.values():
            if work is not None:
                work.wait()
        for st in stages.values():
            with torch.cuda.stream(st["stream"]):
                flush_bucket(st)
                for i in (0, 1):
                    if st["work"][i] is not None:
                        st["work"][i].wait()
                        st["work"][i] = None

    def timed_rounds(run_step, runner=None):
        """The contract's timed region — barrier + synchronize, EXACTLY K steps, synchronize + barrier, MAX over ranks —
        repeated until --min-time seconds have been measured (every rank sees the same maxima, so they stop together).
        Between two rounds the runner is waited for through ITS synchronize() (besides the device-wide one), as a caller that
        works in bursts of K steps would: the runner then starts the next burst's two slots together (pipeline.COMMON_START_US;
        the hold is inside the timed region)."""
        rounds, total, i0, res = [], 0.0, args.warmup, None
        timed_rounds.local = []
        timed_rounds.bad = 0              # OR of the status every round's poll returned (failure codes | shortfall of finished tiles)
        timed_rounds.polls = 0

        def one_round():
            nonlocal i0, res
            rn = runner if runner is not None else cur["runner"]
            if rn is not None:
                # between two rounds (outside the timed region): wait for the runner AND check the previous round — failure
                # codes and the proof of work (workgroup-tiles finished == expected == the host's own count, ops.Workspaces)
                timed_rounds.bad |= rn.poll()
                timed_rounds.polls += 1
            torch.cuda.synchronize()
            gdist.barrier(world)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                res = run_step(i0 + i)
            finish_gathers()                  # every step's all-gather has landed before the clock stops
            torch.cuda.synchronize()
            gdist.barrier(world)
            i0 += args.steps
            local = time.perf_counter() - t0
            timed_rounds.local.append(local)
            return gdist.max_over_ranks(local, dev, world)

        # UNTIMED settling rounds behind the W warm-up steps (more warm-up, same shape as the timed rounds): with the driver's
        # `--warmup 5` the first timed round ran 10-13 % slower than the rest (2.9 ms of work do not bring the clocks up and the
        # first launches still write the CUs' canonical seats), and that one round was the whole 14-16 % "spread" of a line
        # whose other 86 rounds lay within 1 % (p10 .. p90).  Rounds are discarded until two in a row agree within 2 % with
        # the one before (at most 24; every rank sees the same maxima and stops settling together).
        settle = []
        while args.min_time > 0 and len(settle) < args.settle:
            settle.append(one_round())
            if len(settle) >= 3 and all(abs(settle[-k] - settle[-k - 1]) <= 0.02 * settle[-k] for k in (1, 2)):
                break
        timed_rounds.settle = len(settle)
        timed_rounds.settle_rounds = list(settle)     # reported in `timing` (what `value` leaves out is in the line)
        timed_rounds.local = []
        while True:
            dt = one_round()
            rounds.append(dt)
            total += dt
            if total >= args.min_time or len(rounds) >= 1000:
                return rounds, res

    def summarise(rounds):
        r = sorted(rounds)
        med = r[len(r) // 2] if len(r) % 2 else 0.5 * (r[len(r) // 2 - 1] + r[len(r) // 2])
        qt = lambda f: r[min(len(r) - 1, int(f * len(r)))]      # noqa: E731
        return med, {"rounds": len(r), "steps_per_round": args.steps,
                     "round_ms": {"min": round(r[0] * 1e3, 4), "p10": round(qt(0.1) * 1e3, 4), "median": round(med * 1e3, 4),
                                  "p90": round(qt(0.9) * 1e3, 4), "max": round(r[-1] * 1e3, 4)},
                     "spread_pct": round((r[-1] - r[0]) / med * 100, 2), "p10_p90_spread_pct": round((qt(0.9) - qt(0.1)) / med * 100, 2),
                     "untimed_settling_rounds": getattr(timed_rounds, "settle", 0),
                     # the discarded rounds themselves, in time order (--settle 0 times from the first round on): the first one is
                     # the K steps right behind the W warm-up steps, as the bare contract would have timed them
                     "settling_round_us": [int(v * 1e6) for v in getattr(timed_rounds, "settle_rounds", [])],
                     "first_round_after_warmup_ms": (round(getattr(timed_rounds, "settle_rounds", [None])[0] * 1e3, 4)
                                                     if getattr(timed_rounds, "settle_rounds", []) else round(rounds[0] * 1e3, 4)),
                     "round_us": [int(v * 1e6) for v in rounds[:128]]}        # in time order (regimes, drifts)

    gc.collect()
    gc.disable()                     # no collector pauses inside the timed region or the timing pass

    # A cooperative launch that reports a failed inter-workgroup hand-off (bounded spin: its outputs are invalid) voids the
    # measurement it happened in.  At N = 1 that used to end the process; at N = 8 one such launch on one rank would lose the
    # whole scaling line.  Now the RANK it happened on switches, in this process (no exec, no restart), to a degraded form —
    # one step in flight, the placement-independent write-through hand-off in every cooperative launch, the all-gather waited
    # for on the slot's stream before the next launch (no collective kernel beside a staffing window) — and every rank
    # repeats the phase (the collectives of a phase are counted the same on all ranks); the line then carries
    # "degraded": {...} and the per-rank record says which rank.  rc != 0 only if a rank fails in the degraded form too.
    def any_rank(flag):
        return gdist.max_over_ranks(1.0 if flag else 0.0, dev, world) > 0 if world > 1 else bool(flag)

    def degrade(word, phase):
        if cur["degraded"] is not None:
            raise SystemExit(f"rank {rank}: hand-off status {word:#x} in the degraded form too ({phase}): no valid measurement")
        print(f"[bench] rank {rank}: hand-off status {word:#x} during {phase}: switching to the degraded form "
              "(one step in flight, write-through hand-off, serialised collective)", file=sys.stderr, flush=True)
        cur["degraded"] = {"rank": rank, "status": word, "phase": phase, "form": "slots=1, write_through=1, collective serialised"}
        for d in (stages, gathers, last):
            d.clear()
        with gpu_turn(share):
            cur["runner"] = PipelinedRunner(pipe, svc, batch, slots=1, halves=False, write_through=True, auto_degrade=False)

    force_fail = os.environ.get("GNNPN_BENCH_FORCE_DEGRADE")           # test hook: pretend the first status poll of that phase failed
    if force_fail and int(os.environ.get("GNNPN_BENCH_FORCE_DEGRADE_RANK", rank)) != rank:   # ... on that rank only
        force_fail = None
    rounds = idx = R = None
    for attempt in range(3):
        for i in range(args.warmup):
            step(i)
        finish_gathers()
        bad = cur["runner"].poll() if cur["runner"] is not None else 0
        if force_fail == "warmup" and attempt == 0 and cur["runner"] is not None:
            bad |= 0x40
        if os.environ.get("GNNPN_BENCH_DEBUG") and cur["runner"] is not None:
            print("[debug] after warm-up:", bad, [w.placement() for w in cur["runner"].workspaces], file=sys.stderr, flush=True)
        if any_rank(bad):
            if bad:
                degrade(bad, "warm-up")
            continue
        timers.enabled = not args.graph
        rounds, (idx, R) = timed_rounds(step)
        timers.enabled = False
        bad = (cur["runner"].poll() | timed_rounds.bad) if cur["runner"] is not None else 0
        if force_fail == "timed" and attempt == 0 and cur["runner"] is not None:
            bad |= 0x40
        if any_rank(bad):
            if bad:
                degrade(bad, "timed rounds")
            rounds = None
            continue
        break
    if rounds is None:
        raise SystemExit(f"rank {rank}: no clean measurement in three attempts")
    runner = cur["runner"]
    elapsed, timing = summarise(rounds)
    ops.check_status(dev)
    # per-rank record (every rank contributes; rank 0 prints): this rank's own step time, whether it degraded, and the placement
    # counters of its slots' last encoder launches (members placed, seats off their canonical CU, declined seats)
    lr = sorted(timed_rounds.local)
    mine = {"rank": rank, "ms_per_step_local": round(lr[len(lr) // 2] / args.steps * 1e3, 4) if lr else None,
            "degraded": cur["degraded"], "placement_last_launch": [w.placement() for w in runner.workspaces] if runner is not None else None,
            # proof of work at the last poll of the timed rounds (one poll per round; counters are cumulative since the last failure)
            "progress": {"polls": getattr(timed_rounds, "polls", 0), "workspaces": runner.progress()} if runner is not None else None,
            # placement events summed over EVERY cooperative launch of the run (sticky status words 1-2, ABI 9) and per timed step;
            # the all-gather's first and second call on this rank (host wall clock, ms)
            "seats": seat_totals(runner) if runner is not None else None,
            "collective_ms": first_collective_ms,
            "gpu": gpu_identity(0 if share else local_rank)}
    per_rank = [mine]
    if world > 1:
        import torch.distributed as td
        per_rank = [None] * world
        td.all_gather_object(per_rank, mine)
    # self-check: every slot's (overlapped) result equals a single-stream run of the same kernels on the same batch
    decode_impl = runner.decode_impl if runner is not None else 0
    with gpu_turn(share):
        ref = pipe.run(svc, batch, decode_impl=decode_impl)
        torch.cuda.synchronize()
    agreement = None
    if args.precision != "f32":       # agreement of the reduced-precision mode with the f32 path: same batch,
        with gpu_turn(share):
            r32 = ML2PNPipeline(net, low, high, K, precision="f32").run(svc, batch)   # compared on the SELECTED rows (dummy /
            torch.cuda.synchronize()                                  # duplicate candidates are one selection)
        same = (ref["actions"] == r32["actions"]).all(-1)
        agreement = {"problems_with_identical_selection": round(float(same.all(1).float().mean()), 4),
                     "identical_decisions": round(float(same.float().mean()), 5),
                     "mean_abs_R_diff": round(float((ref["R"] - r32["R"]).abs().mean()), 6)}
    if runner is not None:
        for s in range(runner.n_slots):
            if (id(runner), s) not in last:        # fewer steps than slots
                continue
            o = runner.graphs[s].outputs
            with gpu_turn(share):
                rj = pipe.run(svc, batches[last[(id(runner), s)]], decode_impl=decode_impl)
            if not (torch.equal(o["idx_high"], rj["idx_high"]) and torch.equal(o["R"], rj["R"])):
                raise SystemExit(f"slot {s}: overlapped result differs from the single-stream run")
    def kernel_pass(rn):
        """Per-kernel durations of one runner's step: an eager single-stream pass with HIP event pairs around the C-ABI calls
        (events cannot be read back from a replayed graph).  -> (events summary, passes)."""
        timers.events = {}
        for _ in range(3):               # untimed: first eager launches of each kernel in this process, clocks settle
            rn.reference_run(0)
        torch.cuda.synchronize()
        # An event pair measures GPU time between its two markers, so a host stall between them would be
        # booked as kernel time once the stream has run dry (a generation-2 Python GC pass over the
        # synthetic dataset's objects costs 30-70 ms and used to land inside one pointer_decode call per
        # pass: the collector is off from before the warm-up to here).  Syncing every 4 passes bounds how
        # far the host runs ahead; the kernels timed here all follow >=0.3 ms of queued work.
        timers.enabled = True
        n = min(args.steps, 12)
        for i in range(n):
            rn.reference_run(0)
            if i % 4 == 3:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        timers.enabled = False
        return timers.summary(), n

    def kernel_table(summary, n_timed, precision):
        out = []
        for name, (avg_ms, n) in sorted(summary.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            c = algorithmic_cost(name, w, B, precision)
            achieved = c["work"] / (avg_ms * 1e-3) / (1e12 if c["bound"] == "mfma" else 1e9)
            k = {"kernel": name, "launches_per_step": n // n_timed, "avg_ms": round(avg_ms, 4), "bound": c["bound"],
                 "achieved": round(achieved, 3), "peak": c["peak"], "unit": c["unit"], "frac": round(achieved / c["peak"], 5)}
            if "fp32_equivalent_work" in c:      # the algorithmic fp32 flops of the product the six fp16 products reproduce
                eq = c["fp32_equivalent_work"] / (avg_ms * 1e-3) / 1e12
                k["fp32_equivalent"] = {"achieved": round(eq, 3), "unit": "TFLOP/s", "vs_fp32_matrix_peak": round(eq / PEAK_F32_TFLOPS, 5)}
            out.append(k)
        return out

    n_timed, summary = args.steps, timers.summary()
    if args.graph and not args.no_kernel_timers:
        with gpu_turn(share):
            summary, n_timed = kernel_pass(runner)
    gc.enable()

    # Second measurement, reported beside the headline and never as `value`: the same workload in the OTHER arithmetic of the
    # recurrent products (f32 <-> exact split), with its own kernel table and the agreement of the two results.
    other_line = None
    other = {"split": "f32", "f32": "split"}.get(args.precision)
    PREC_TEXT = {"f32": "W_hh.h of encoder and decoder on the fp32 matrix cores (v_mfma_f32_16x16x4_f32)",
                 "split": "W_hh.h of encoder and decoder from fp32 operands split EXACTLY into three fp16 pieces, six fp16 "
                          "products per term (every cross term >= 2^-24 kept), fp32 accumulate; rest f32"}
    if other and args.graph and not args.no_split_line:
        pipe_s = ML2PNPipeline(net, low, high, K, precision=other)
        with gpu_turn(share):
            runner_s = PipelinedRunner(pipe_s, svc, batch, slots=n_slots, halves=False if share else None)
        gc.collect()
        gc.disable()
        for i in range(args.warmup):
            step(i, runner_s)
        rounds_s, _ = timed_rounds(lambda i: step(i, runner_s), runner_s)
        el_s, timing_s = summarise(rounds_s)
        bad_s = runner_s.poll()                    # the second line is not worth the first: a failed hand-off here is recorded, not fatal
        out_s = runner_s.graphs[0].outputs
        with gpu_turn(share):
            ref_s = pipe.run(svc, batches[last[(id(runner_s), 0)]], decode_impl=decode_impl)
        same = (out_s["actions"] == ref_s["actions"]).all(-1)
        other_line = {"precision": other, "arithmetic": PREC_TEXT[other],
                      **({"INVALID_hand_off_status": bad_s} if bad_s else {}),
                      "value": round(world * B * args.steps / el_s, 2), "unit": "problems/s",
                      "ms_per_step": round(el_s / args.steps * 1e3, 4), "timing": timing_s,
                      f"agreement_vs_{args.precision}": {
                          "problems_with_identical_selection": round(float(same.all(1).float().mean()), 4),
                          "identical_decisions": round(float(same.float().mean()), 5),
                          "max_abs_R_diff": round(float((out_s["R"] - ref_s["R"]).abs().max()), 6)}}
        if not args.no_kernel_timers:
            with gpu_turn(share):
                sm, nt = kernel_pass(runner_s)
            other_line["kernels"] = kernel_table(sm, nt, other)
        gc.enable()
        del runner_s

    if rank != 0:
        gdist.destroy(world)
        return
    if use_dist and bucket == 1 and tuple(idx.shape) != (world * B, T):
        raise SystemExit(f"all-gather returned {tuple(idx.shape)}, expected {(world * B, T)}")
    if use_dist and bucket > 1:
        for st in stages.values():
            if st.get("gathered_rows", 0) % (world * B) != 0 or st.get("gathered_rows", 0) == 0:
                raise SystemExit(f"bucketed all-gather returned {st.get('gathered_rows')} rows, expected a multiple of {world * B}")
    value = world * B * args.steps / elapsed
    kernels = kernel_table(summary, n_timed, args.precision)
    # HBM traffic per launch from the committed PMC passes (rocprofv3 cannot run inside this process):
    # profiles/r05_pmc_traffic.json (tools/r05_profiles.sh + tools/collect_profiles.py; r04's while that one is absent), FETCH_SIZE doubled as the gfx950 guide
    # prescribes; only quoted when this run is a workload / batch / precision those passes measured.
    # Both summaries carry the source hash of the tree they were measured on (gnnpn_sc_amd._lib.source_hash); they are quoted only
    # while it equals the loaded library's — after any kernel change they read null until the passes have been re-run.
    from gnnpn_sc_amd import _lib as _glib
    lib_hash = _glib.source_hash()
    counters_state = {}

    def committed(name):
        """A committed counter summary, or None when it was not measured on the sources of the kernels it is about (`source_group`:
        the recurrent kernels' or the aggregate's files, gnnpn_sc_amd._lib.SOURCE_GROUPS; absent: the whole tree)."""
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
        now = _glib.source_hash(d.get("source_group"))
        counters_state[name] = "current" if d.get("source_hash") == now else f"stale (measured on sources {d.get('source_hash')}, the tree's are {now}): not quoted"
        return d if d.get("source_hash") == now else None

    def pmc_traffic(precision):
        try:
            name = next(n for n in ("r06_pmc_traffic.json", "r05_pmc_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
            pmc = committed(name)
            if pmc is None:
                return {}
            # the summary's hash covers the recurrent kernels' sources; the other kernels of the table are quoted only while the WHOLE tree is unchanged
            rest_ok = pmc.get("source_hash_all") == lib_hash
            return {k: v["traffic"] for k, v in pmc.get(f"{args.workload}_b{B}_{precision}", {}).get("kernels", {}).items()
                    if rest_ok or k in ("lstm_encode", "pointer_decode")}
        except (OSError, ValueError, KeyError, StopIteration):
            return {}
    # Matrix-pipe utilisation of the two recurrent kernels (BASELINE north_star: "MFMA utilisation ... against gfx950 peak") from the
    # committed SQ-counter passes of the same eager command (tools/r05_recurrent_sq.sh -> profiles/r05_recurrent_sq_summary.json):
    # SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).  Quoted, like `traffic`: rocprofv3 cannot run inside this process.
    def sq_mfma_util(precision):
        try:
            name = next(n for n in ("r06_recurrent_sq_summary.json", "r05_recurrent_sq_summary.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
            sq = committed(name)
            if sq is None:
                return {}
            sq = sq["kernels"]
            names = {"lstm_encode": "lstm_encode_coop_kernel", "pointer_decode": "pointer_decode_lean_kernel"}
            return {k: sq[f"{args.workload}/{precision}/{n}"]["derived"]["mfma_busy_frac_of_all_simds"] for k, n in names.items()
                    if f"{args.workload}/{precision}/{n}" in sq and B == WORKLOADS[args.workload]["B"]}     # the passes ran the default batch
        except (OSError, ValueError, KeyError, StopIteration):
            return {}
    traffic = pmc_traffic(args.precision)
    util = sq_mfma_util(args.precision)
    for k in kernels:
        k["traffic"] = traffic.get(k["kernel"])
        if k["kernel"] in util:
            k["mfma_util"] = util[k["kernel"]]
    if other_line is not None and "kernels" in other_line:
        t2, u2 = pmc_traffic(other), sq_mfma_util(other)
        for k in other_line["kernels"]:
            k["traffic"] = t2.get(k["kernel"])
            if k["kernel"] in u2:
                k["mfma_util"] = u2[k["kernel"]]
    if world == 1 and args.graph and not args.no_kernel_timers:
        agg = batched_aggregate_roofline(table, B, dev)
        try:     # the committed PMC passes of tools/r04_aggregate_profiles.sh, keyed by shape (rows per copy x copies) and kernel
            pname = next(n for n in ("r06_csr_aggregate_pmc_traffic.json", "r05_csr_aggregate_pmc_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
            pm = committed(pname)
            if pm is None:
                raise KeyError(pname)
            kname = {"tiled": "csr_aggregate_tiled_kernel", "l2-gather": "csr_aggregate_kernel"}.get(agg["form"])
            agg["traffic"] = pm["shapes"][f"{agg['rows'] // agg['copies']}x{agg['copies']}"][kname]["traffic"]
        except (OSError, ValueError, KeyError, ZeroDivisionError, StopIteration):
            agg["traffic"] = None
    roof = None
    if kernels:
        k0 = kernels[0]
        roof = {"kernel": k0["kernel"], "bound": k0["bound"], "achieved": k0["achieved"], "peak": k0["peak"],
                "unit": k0["unit"], "frac": k0["frac"], "traffic": k0["traffic"],
                # where `traffic` and `mfma_util` come from: committed rocprofv3 --pmc passes of the same eager command, NOT this run
                "counters_from": {"what": "committed rocprofv3 --pmc passes of the same eager command: FETCH_SIZE x 2 + WRITE_SIZE per launch; SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles",
                                  "library_source_hash": lib_hash, "files": counters_state}}
        if "mfma_util" in k0:
            roof["mfma_util"] = k0["mfma_util"]
        if "fp32_equivalent" in k0:      # split: `achieved` counts the six EXECUTED f16 products per fp32 term
            roof["fp32_equivalent"] = k0["fp32_equivalent"]
    line = {
        "metric": "service-composition problems/sec (ML+2PN inference)", "value": round(value, 2),
        "unit": "problems/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": {"f32": "f32", "f16": "f16 encoder operands / f32 rest",
                                       # its own token, not "f32": a consumer keyed on dtype must not compare this line with an
                                       # fp32-MFMA line of another round as like with like; `agreement_vs_f32` is always in the line
                                       "split": "f32-split3xf16 (f32 operands and results; recurrent W_hh.h products: fp32 operands split exactly into 3 x f16, "
                                                "six f16-matrix-core products per term, f32 accumulate; everything else f32 arithmetic)"}[args.precision],
        "precision": args.precision,
        "data": "synthetic", "timing": timing,
        "config": {"workload": w["desc"], "batch_per_gpu": B, "global_batch": B * world,
                   "resident_batches": len(batches),
                   "launch": ("eager, one stream" if runner is None else
                              "hipGraph replay, one step in flight, its recurrent part as two half-batches side by side on two HIP "
                              "streams (cooperative launches paired on every CU)" if runner.halves else
                              f"hipGraph replay, {runner.n_streams} independent step(s) in flight on separate HIP streams"),
                   "kernel_timing": ("HIP events in a separate single-stream pass" if args.graph else
                                     "HIP events inside the timed region (durations include overlap with the "
                                     "other in-flight step)"),
                   "service_embedding": "problem-independent GCN branch evaluated once per (weights, service table), "
                                        "outside the step (SURVEY.md section 7)",
                   "weights": "random-init (PyTorch defaults, seed 0)", "parallelism": f"dp{world}",
                   "collective": (None if not use_dist else "one asynchronous all-gather of the selected indices per step" if bucket == 1 else
                                  f"one asynchronous all-gather per {bucket} steps of a slot (the selected indices of those steps, staged on the device)"),
                   **({"slot_stream_priority": runner.stream_priority} if runner is not None else {}),
                   **({"rank0_cpu_affinity": affinity} if affinity is not None else {}),
                   **({"NOT_A_MEASUREMENT": "GNNPN_BENCH_SHARE_GPU=1: all ranks share one GPU over gloo (launch-path check)"}
                      if share else {})},
        "roofline": roof, "kernels": kernels + ([agg] if world == 1 and args.graph and not args.no_kernel_timers else []),
    }
    if world == 1 and args.graph and not args.no_kernel_timers and not share and not use_dist:
        with gpu_turn(share):
            n_pcie = max(100, min(2000, int(0.25 / max(elapsed / args.steps, 1e-6))))      # about a quarter of a second of steps
            pv = pcie_inclusive_rate(runner, batches, B, T, n_pcie, dev)
        if runner.poll() == 0:
            in_b = sum(t.numel() * t.element_size() for t in PipelinedRunner._fields(batches[0]))
            line["pcie_inclusive"] = {"value": round(pv, 2), "unit": "problems/s", "ratio_to_value": round(pv / value, 4), "steps": n_pcie,
                                      "per_step": f"one host-to-device copy of {in_b // 1024} KB (pinned arena in the slots' layout), "
                                                  f"one device-to-host copy of {(B * T + B) * 4 // 1024} KB (idx_high, R)",
                                      "note": "never `value`: the timed region of `value` starts with the inputs resident in HBM"}
    line["per_rank"] = per_rank
    if any(r.get("degraded") for r in per_rank if r):
        line["degraded"] = [r["degraded"] for r in per_rank if r and r.get("degraded")]
    if agreement is not None:
        line["agreement_vs_f32"] = agreement
    if other_line is not None:
        line["other_precision"] = other_line
    if cpu_line is not None:
        line["cpu_baseline"] = cpu_line
    try:                   # whatever the libraries buffered on the C stdout goes where fd 1 points now: stderr
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001
        pass
    sys.stdout.flush()
    os.write(REAL_STDOUT, (json.dumps(line) + "\n").encode())
    gdist.destroy(2 if force_dist else world)


if __name__ == "__main__":
    main()
