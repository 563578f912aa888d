"""Device-resident end-to-end ML+2PN inference: what the reference spreads over
``TrainML.test`` (/root/reference/src/models/trainML.py:49-72) -> JSON -> ``loadDataPN``
(src/loadData.py:72-152) -> JSON -> the eval block of ``TrainModel.train_and_validate``
(src/models/trainPNHigh.py:131-150), as one stream of kernels with nothing leaving HBM between
the stages:

    GNN scores [B,S] -> per-category top-K feasible candidates -> PN input rows [B,L,8]
    -> Low/High encoders (one launch) -> Low decode -> High decode -> QoS reward

Inputs are ``DeviceBatch``/``DeviceServices`` (packed once from the host structures of synth.py /
loadData.py); outputs stay on the device.
"""
import ctypes
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib, custom_ops, graph, ops   # noqa: F401  (custom_ops registers torch.ops.gnnpn.*)
from .modelPN import two_level_greedy


@dataclass
class DeviceServices:
    """Problem-independent service side, resident in HBM."""
    x_service: torch.Tensor      # f32 [S,5]
    csr: graph.CSR               # self-loop-complete destination-major CSR, raw weights
    cat_ptr: torch.Tensor        # i32 [T+1]
    qos: torch.Tensor            # f64 [S,4]
    emb_cache: tuple = None      # (prepared-weights object, [S,hidden] service embedding): see ML2PNPipeline.service_embedding

    @staticmethod
    def from_table(table, device):
        ei = torch.from_numpy(np.ascontiguousarray(table.edge_index))
        ea = torch.from_numpy(np.ascontiguousarray(table.edge_attr))
        csr = graph.gcn_csr(ei, ea, table.n_services).to(device)
        return DeviceServices(torch.from_numpy(table.x_service).to(device), csr,
                              torch.from_numpy(np.ascontiguousarray(table.cat_ptr)).to(device),
                              torch.from_numpy(np.ascontiguousarray(table.qos)).to(device))


@dataclass
class DeviceBatch:
    """B composition requests, resident in HBM."""
    x: torch.Tensor              # f32 [N,7]
    wf_csr: graph.CSR
    seg_ptr: torch.Tensor        # i32 [B+1]
    local_bounds: torch.Tensor   # f64 [B,T,4]
    present: torch.Tensor        # u8  [B,T]
    global_bounds: torch.Tensor  # f64 [B,4]
    max_nodes: int = 0           # > 0: every workflow graph has at most this many nodes AND every edge stays inside its
    #                              graph (host-checked when the batch was packed) -> the one-launch GIN branch may be used

    @property
    def n_problems(self):
        return self.present.shape[0]

    @staticmethod
    def from_problems(pb, device):
        ei = torch.from_numpy(np.ascontiguousarray(pb.edge_index))
        batch = torch.from_numpy(np.ascontiguousarray(pb.batch))
        n = pb.x.shape[0]
        inside = bool((pb.batch[pb.edge_index[0]] == pb.batch[pb.edge_index[1]]).all()) if pb.edge_index.size else True
        max_nodes = int(np.bincount(pb.batch, minlength=1).max()) if inside and n else 0
        return DeviceBatch(torch.from_numpy(pb.x).to(device), graph.csr_by_destination(ei, n).to(device),
                           graph.segment_ptr(batch, pb.n_problems).to(device),
                           torch.from_numpy(np.ascontiguousarray(pb.local_bounds)).to(device),
                           torch.from_numpy(np.ascontiguousarray(pb.present)).to(device),
                           torch.from_numpy(np.ascontiguousarray(pb.global_bounds)).to(device), max_nodes)

    def shard(self, rank, world):
        """Contiguous shard of the problems for one rank (dist.py); graphs stay whole."""
        B = self.n_problems
        lo, hi = B * rank // world, B * (rank + 1) // world
        seg = self.seg_ptr.cpu()
        n0, n1 = int(seg[lo]), int(seg[hi])
        rp = self.wf_csr.rowptr.cpu()
        e0, e1 = int(rp[n0]), int(rp[n1])
        dev = self.x.device
        csr = graph.CSR((rp[n0:n1 + 1] - e0).to(dev), (self.wf_csr.col[e0:e1] - n0).contiguous(), None, n1 - n0)
        return DeviceBatch(self.x[n0:n1].contiguous(), csr, (seg[lo:hi + 1] - n0).to(dev),
                           self.local_bounds[lo:hi].contiguous(), self.present[lo:hi].contiguous(),
                           self.global_bounds[lo:hi].contiguous(), self.max_nodes)


# Hand-off form of the cooperative kernels in the runner's launches (gnnpn_launch_opts_t.write_through): False = granule stores
# that stay in the group's XCD L2 (faster; correct by this toolchain's lowering and by the run-time XCD placement of a group),
# True = agent-scope write-through stores (valid by the HIP memory model, placement independent).  What can go wrong with the
# fast form is liveness, not data: a granule is ONE 8-byte store carrying its own tag, so a reader sees an old granule or a new
# one, never a mixture — a store that does not become visible ends in a bounded-spin time-out (status 1 / 2), and a launch that
# did not run at all in a shortfall of finished workgroup-tiles (status 16): both are loud at the next poll / synchronize, and
# the runner then switches ITSELF to the write-through form for the rest of its life (``auto_degrade``).  DESIGN.md sections 4.5-4.6.
DEFAULT_WRITE_THROUGH = os.environ.get("GNNPN_PIPE_WRITE_THROUGH", "0") == "1"
HOST_COPY_ON_ITS_OWN_STREAM = os.environ.get("GNNPN_HOST_COPY_INLINE") != "1"
COMMON_START_US = 0.0       # PipelinedRunner: > 0 holds the first replays of a burst until both are enqueued, at most this long.  OPT-IN since
#                             round 5 (GNNPN_PIPE_COMMON_START_US=400 is what round 4 ran): the slots no longer slip apart once the front-half
#                             kernels share the cooperative kernels' LDS footprint (FRONT_LDS_KB below), which is also 1 % faster than the gate


def half_batch_split(n_problems):
    """Where a batch is cut for the two half-batches that run side by side: whole tiles of 16 problems to the first half,
    the (possibly ragged) rest to the second; 0 when there is nothing to put on the second stream."""
    half = ((int(n_problems) + 31) // 32) * 16
    return half if 0 < half < n_problems else 0


_warned_partitioned = set()


def warn_partitioned_device(device, precision):
    """Say ONCE per device what a compute-partitioned GPU means for this path.  The cooperative recurrent
    kernels — the ones every measured number of this package comes from — are placed for 8 XCDs x 32 CUs (MI355X in SPX mode:
    csrc/coop_common.h::coop_place deals seats per XCD and assumes eight of them).  On a device that shows fewer than 256 compute
    units (CPX / DPX / QPX partitions, or another GPU) their launchers return GNNPN_E_UNSUP: with precision "split" or "f16" the
    recurrent calls then RAISE; with "f32" the decoder in auto mode (impl 0) takes the per-workgroup streaming form and the encoder
    raises unless impl=1 is passed — the streaming forms re-stream the 1 MiB W_hh from L2 every step, about 10 x slower per step
    (11 us against 2.4 us / 3.2 us at the QWS shape, csrc/lstm_coop.hip)."""
    import warnings
    if device.type != "cuda":
        return
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _warned_partitioned:
        return
    _warned_partitioned.add(idx)
    n_cu = torch.cuda.get_device_properties(idx).multi_processor_count
    if n_cu < 256:
        warnings.warn(f"gnnpn: cuda:{idx} shows {n_cu} compute units; the cooperative recurrent kernels are built for 8 XCDs x 32 CUs (MI355X, SPX "
                      f"mode). precision={precision!r}: " + ("the recurrent launches will raise GNNPN_E_UNSUP — " if precision != "f32" else "") +
                      "only the per-workgroup streaming forms (precision='f32', impl=1) run here, about 10 x slower per recurrent step.",
                      RuntimeWarning, stacklevel=3)


class ML2PNPipeline:
    """net: modelML.Net; low/high: modelPN.CombinatorialRL (levels "Low"/"High")."""

    def __init__(self, net, low, high, n_per, precision=None):
        from .modelPN import default_precision
        self.net, self.low, self.high, self.n_per = net, low, high, n_per
        # None: modelPN.default_precision — the exact split wherever its kernels apply (what bench.py measures), else fp32;
        # "f16": opt-in fp16-operand encoder (not parity-exact)
        self.precision = default_precision(low, high) if precision is None else precision
        self.cache_service_embedding = True   # False: re-evaluate the GCN branch in every pass (round-1 behaviour)
        self._side_streams = {}
        warn_partitioned_device(next(low.parameters()).device, self.precision)

    @torch.no_grad()
    def service_embedding(self, services):
        """The GCN (service) branch of Net.forward (modelML.py:145-156,164): a function of the weights and the service
        table ONLY, so it is evaluated once per (weights, table) and kept on the table ("compute
        the service embedding once per model and cache it") instead of once per batch.  Loading / moving the weights
        gives a new prepared-weights object and so invalidates the entry."""
        prep = self.net.prepared(services.x_service.device)
        c = services.emb_cache
        if c is None or c[0] is not prep:
            c = services.emb_cache = (prep, self.net.service_embedding(services.x_service, services.csr))
        return c[1]

    @torch.no_grad()
    def scores(self, services, batch):
        """TrainML.test's forward (trainML.py:56-58)."""
        return self.net.scores(batch.x, batch.wf_csr, batch.seg_ptr, services.x_service, services.csr,
                               service_emb=self.service_embedding(services) if self.cache_service_embedding else None,
                               max_nodes=batch.max_nodes, dense_precision="split" if self.precision == "split" else "f32")

    @torch.no_grad()
    def candidates(self, services, batch, scores):
        """sort + loadDataPN + SCDataset, fused (trainML.py:62; loadData.py:101-149; trainPNHigh.py:23-31)."""
        return torch.ops.gnnpn.segment_topk_feasible(scores, services.cat_ptr, services.qos, batch.local_bounds,
                                                     batch.present, batch.global_bounds, self.n_per)

    @torch.no_grad()
    def run(self, services, batch, decode_impl=0, lds_kb=0, ws=None, paired_start=False, write_through=False):
        """One pass.  decode_impl / lds_kb / paired_start / ws: launch options of the recurrent kernels (modelPN.two_level_greedy;
        paired_start: the caller starts these launches together with a partner's — half-batches do by construction).
        ``ws`` = a PAIR of workspaces: the recurrent part runs as two half-batches side by side, the second on a side
        stream (fork / join, capturable) — the problems are independent, and two cooperative launches of one workgroup
        per CU each share every CU for their whole length (encoder beside encoder, decoder beside decoder), which two
        whole batches in flight on two slots do only where their timelines happen to line up."""
        scores = self.scores(services, batch)
        rows, ids = self.candidates(services, batch, scores)
        if isinstance(ws, (tuple, list)):
            B = rows.shape[0]
            half = half_batch_split(B)
            if len(ws) != 2 or not half:
                raise ops.GnnpnError(f"ML2PNPipeline.run: two half-batches need 2 workspaces and more than 16 problems (got {len(ws)}, {B})")
            cur = torch.cuda.current_stream(rows.device)
            side = self._side_streams.get(rows.device)
            if side is None:
                side = self._side_streams[rows.device] = torch.cuda.Stream(rows.device)
            # everything both halves share is produced on the CURRENT stream before the fork: the packed / folded weights
            # (pack kernels and host-to-device copies of a first call) and the lazily allocated workspaces would otherwise be
            # issued on the side stream by the half that runs first in Python, with nothing ordering the other half's
            # reads behind them
            for net in (self.low, self.high):
                net.actor.packed()
                net.actor.check_precision(self.precision)
            n_cat = rows.shape[1] // self.n_per
            for w in ws:
                w.encode()
                w.decode(max(half, B - half), n_cat, self.n_per)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                out_b = two_level_greedy(self.low, self.high, rows[half:], precision=self.precision, decode_impl=decode_impl,
                                         lds_kb=lds_kb, ws=ws[1], paired_start=True, write_through=write_through)
            out_a = two_level_greedy(self.low, self.high, rows[:half], precision=self.precision, decode_impl=decode_impl,
                                     lds_kb=lds_kb, ws=ws[0], paired_start=True, write_through=write_through)
            cur.wait_stream(side)
            for v in out_b.values():
                v.record_stream(cur)
            out = {k: torch.cat([out_a[k], out_b[k]]) for k in out_a}
        else:
            out = two_level_greedy(self.low, self.high, rows, precision=self.precision, decode_impl=decode_impl,
                                   lds_kb=lds_kb, ws=ws, paired_start=paired_start, write_through=write_through)
        out.update(scores=scores, pn_inputs=rows, candidate_ids=ids)
        return out

    def capture(self, services, batch, warmup=2, decode_impl=0, lds_kb=0, ws=None, paired_start=False, pool=None, write_through=False):
        """Record one whole pass over (services, batch) into a HIP graph and return a callable that
        replays it on the CURRENT stream (one launch per step instead of ~25).  The returned dict's
        tensors are the graph's static outputs: they are overwritten by every replay.  ``ws`` is the
        private ``ops.Workspaces`` of this graph (default: a new one), so that graphs may be in flight
        at the same time on different streams (independent batches pipelined); it is frozen — the graph
        holds its addresses — and lives as long as the returned callable.  ``pool``: a graph memory pool to capture
        into (torch.cuda.graph's argument; NOT for graphs whose outputs must survive each other's replays)."""
        ws = ops.new_workspaces(batch.x.device) if ws is None else ws
        all_ws = list(ws) if isinstance(ws, (tuple, list)) else [ws]
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            for _ in range(warmup):          # allocate workspaces / pack weights outside the capture
                self.run(services, batch, decode_impl, lds_kb, ws, paired_start, write_through)
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # thread_local: only THIS thread's calls are policed during the capture.  In the default ("global") mode an event query
        # made by another thread while the capture is open is an error that invalidates it — and the watchdog thread of
        # torch.distributed's RCCL process group queries the events of its collectives whenever one is outstanding: a runner
        # captured while collectives are in flight (bench.py's degraded form at N > 1 is built mid-run) died that way in one
        # of this round's GPU runs (ProcessGroupNCCL watchdog: HIP error from hipEventQuery, the process aborted)
        with torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local"):
            out = self.run(services, batch, decode_impl, lds_kb, ws, paired_start, write_through)
        for w in all_ws:
            w.frozen = True
        booked_replay = ops.graph_replay(graph, all_ws)      # graph.replay() + the host's count of the cooperative launches it makes

        def replay():
            booked_replay()
            return out
        replay.graph, replay.outputs, replay.workspaces = graph, out, ws
        return replay

    @torch.no_grad()
    def rankings(self, services, batch):
        """The artefact TrainML.test writes (trainML.py:62-68,148-149): full ranking per problem."""
        return torch.ops.gnnpn.rank_rows(self.scores(services, batch))

    @torch.no_grad()
    def test(self, services, batch, labels):
        """TrainML.test (trainML.py:49-72): (idxList = full ranking per problem, [P@1, P@5]).
        ``labels`` [B,S] 0/1 (float32 device tensor).  Rankings stay on the device (int32 [B,S])."""
        ranking = self.rankings(services, batch)
        pk = torch.ops.gnnpn.precision_at_k(ranking, labels.float().contiguous(), [1, 5])
        return ranking, [float(v) for v in pk.mean(0).tolist()]


# device index -> weak reference to the PipelinedRunner whose replays were enqueued last on that device (PipelinedRunner._take_turn)
_last_runner_on_device = {}


def _has_collective_stream():
    """True in a rank of an RCCL ("nccl") process group: its collectives run on a stream of the process group's own."""
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()):
        return False
    try:
        return "nccl" in str(td.get_backend()).lower()
    except Exception:       # a backend object without a name: assume it brings a stream
        return True


class PipelinedRunner:
    """Throughput mode: ``slots`` independent batches in flight on ``slots`` HIP streams — or, for batches of 512 problems
    and more, ONE batch in flight whose recurrent part runs as two half-batches side by side (``halves``: the ``slots``
    static input / output sets then take turns on one stream); two slots with recurrences of 2000 steps and more start
    their replays in pairs (``lockstep``).  In every mode ``n_slots`` is the reuse distance of ``submit``'s outputs.
    See __init__.

    The recurrent kernels are step-latency-bound and leave most of the machine idle, so consecutive
    (independent) batches are overlapped: every slot owns one captured HIP graph of the whole pass, its
    own static input/output tensors and its own cooperative-kernel hand-off workspaces; batch i runs on
    slot i % slots.  With two slots the 16-member decoder form is selected (256 registers: it shares
    every SIMD with a wave of the other slot's encoder).  All batches must have the shapes of
    ``example_batch`` (graphs are shape-static); ``submit`` copies a new batch into the slot's static
    tensors on the slot's stream (``batch=None`` re-runs the resident one, as bench.py does).
    """

    def __init__(self, pipe, services, example_batch, slots=2, halves=None, write_through=None, auto_degrade=True,
                 stream_priority=None):
        # Batches of 512 problems and more: ONE batch in flight, its recurrent part as two half-batches side by side
        # (ML2PNPipeline.run with a pair of workspaces).  A cooperative launch has one workgroup per CU and two of them
        # fill a CU's registers, so nothing else runs beside a co-resident pair; with two WHOLE batches in flight on two
        # slots the pairing is left to chance — one slot's front half waits for the other's 25 ms encoder, encoders meet
        # decoders — and the 1000-task shape ran 28.3 ms per batch where its recurrent kernels, always paired, need 21.1
        # (tools/bench_slot_parts.py).  Below 512 problems a half-batch launch no longer fills the chip: two slots.
        n = example_batch.n_problems
        self.halves = bool(halves) if halves is not None else (n >= 512 and int(slots) > 1 and
                                                               os.environ.get("GNNPN_PIPE_HALVES", "1") != "0")
        # ``n_slots`` = the number of static input / output sets = the reuse distance of submit()'s outputs, in EVERY mode
        # (the half-batch mode used to collapse to one slot, and a caller following the documented rule read
        # outputs the next replay was already overwriting).  ``n_streams`` = steps in flight: the half-batch mode keeps ONE
        # step in flight, so its slots' graphs replay one after the other on one stream.
        self.pipe, self.services, self.n_slots = pipe, services, max(1, int(slots))
        # ... and never more than TWO: a CU holds two cooperative workgroups, so two replays in flight are what the chip has room for —
        # a third free-running stream put three launches in front of those two slots and ended in half-staffed launches and
        # bounded-wait time-outs (round 6, tools/probes/dbg_three_slots.py: slots=3 ran 24 k problems/s with status 0x13).  More slots
        # than two are more static input / output sets (a longer reuse distance), dealt onto the two streams in turn.
        self.n_streams = 1 if self.halves else min(self.n_slots, 2)
        # Stream priority of the slots: HIP deals a process's streams onto a few in-order hardware queues (GPU_MAX_HW_QUEUES,
        # default 4) round robin, PER PRIORITY.  In a rank of an RCCL process group the collective's stream waits for one slot's
        # step; when that wait shares a hardware queue with the other slot's stream it sits in front of that slot's launches
        # (forced world-1 all-gather at the QWS shape: 449 k problems/s against 480 k without a process group).  Slot streams of a
        # priority of their own never share a queue with the process group's (normal-priority) stream: 470 k, the same as 8
        # hardware queues give (467 k) but without an environment variable that has to be set before the runtime starts — and
        # the two do NOT add up (8 queues AND high priority: 367-378 k).  Without a process group high priority costs 0.35 %
        # (477.8-478.2 k against 479.5-479.7 k), so it is used only where a collective stream exists.  profiles/LOG_r06.md §9.
        self._priority_chosen = stream_priority is None and os.environ.get("GNNPN_PIPE_STREAM_PRIORITY") is None
        if stream_priority is None:
            env = os.environ.get("GNNPN_PIPE_STREAM_PRIORITY")
            # (a process started with 8 or more hardware queues has the separation already — and must not get both)
            many_queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) >= 8
            # Only for free-running slots on streams of their own: the half-batch mode keeps ONE step in flight on one stream (nothing of
            # another slot for the collective's wait to sit in front of), and there high priority COSTS — Normal shape, forced RCCL:
            # 199 k problems/s against 234-238 k at normal priority (238 k without a process group); 1000-task shape 22.8 k against 23.8 k
            # (tools/r06/dist_priority_shapes.sh)
            free_running = self.n_streams > 1
            stream_priority = int(env) if env is not None else (-1 if free_running and _has_collective_stream() and not many_queues else 0)
        self.stream_priority = int(stream_priority)
        self.streams = [torch.cuda.Stream(priority=self.stream_priority) for _ in range(self.n_streams)]
        # decoder form beside another slot's kernels: the 8-member build sized for two workgroups per CU (decode_impl 4).  Measured at QWS B=256 against the 16-member form (3):
        # 228 k vs 221 k problems/s in fp32, 353 k vs 316 k with the split precision.
        shared = 4
        paired = self.n_streams > 1 or self.halves    # two cooperative launches share every CU
        self.decode_impl = int(os.environ.get("GNNPN_PIPE_DECODE_IMPL", shared if paired else 0))
        # Placement: the cooperative kernels claim one CU per workgroup at run time (csrc/coop_common.h, coop_place), so
        # the two slots' launches share every CU one workgroup each whatever the dispatcher does; the LDS-footprint
        # padding round 1 steered the dispatcher with (100 / 56 KB) is no longer needed and stays as an option only.
        # Exact-split precision: the recurrent kernels hold the third weight piece in LDS (60 KB encoder, 69 KB decoder).  Two
        # such workgroups fit a CU's 160 KB, but LDS is allocated in CONTIGUOUS ranges: with unequal footprints a freed 60 KB
        # hole does not take the next 69 KB workgroup while the neighbour's range sits in the middle, the launch stays
        # under-staffed, and two launches can starve each other until the bounded spins give up (seen: bench.py, precision
        # split, two slots: members missing at the first sweep, 20 of 32 seats of an XCD staffed after 0.3 s).  Every
        # cooperative launch of a multi-slot runner is therefore padded to ONE footprint, 78 KB: any freed range fits any
        # waiting workgroup.  (f32: 19 / 27 KB — nothing to equalise.)
        env = os.environ.get("GNNPN_SLOT_LDS_KB")
        equal = 78 if (getattr(pipe, "precision", "f32") == "split" and paired) else 0
        self.lds_kb = [int(v) for v in env.split(",")] if env else [equal] * self.n_slots
        # ... and so are the ORDINARY kernels of a step that use LDS in front of the encoder (the one-launch GIN branch, the score
        # product: ops.lds_footprint / gnnpn_lds_footprint_kb).  A cooperative workgroup that lands above such a kernel's few KB
        # keeps its 78 KB in the middle of the CU when the small kernel has gone, and the OTHER slot's encoder workgroup for that
        # CU finds no contiguous 78 KB until this encoder has finished: the two free-running slots slipped apart by 0.5 ms in
        # every 20-step round (round 4 held the slots of a burst behind a common-start gate against it: 455 k problems/s with 4-8
        # slow rounds of 89; with the front half at the same footprint 460 k and no slow round in 270, gate off —
        # profiles/r05_slip_front_lds.jsonl).  Two free-running slots only: the half-batch mode runs its front half alone.
        env = os.environ.get("GNNPN_PIPE_FRONT_LDS_KB")
        self.front_lds_kb = int(env) if env is not None else (equal if (self.n_streams == 2 and not self.halves) else 0)
        self.count = 0
        # Two slots, long recurrent kernels: start the two replays of a pair TOGETHER (a submission joins the leader that is
        # still waiting for a partner, else it leads).  Free-running slots drift apart by the difference of their step
        # times, and a cooperative launch that starts under the OTHER slot's front half finds every CU's LDS occupied by
        # short-lived neighbours: its workgroups land above them (coop_place gives such seats back, up to a few hundred times
        # per launch), seats are taken off their canonical CUs, and from there on both slots' launches run at the speed of
        # one — the 2000-task shape alternated between 41 and 62 ms per step (spread 37-48 % over rounds; tools/slot_overlap.py).
        # Started together, the two front halves run side by side and are gone when the encoders arrive.  The wait costs the
        # tail by which the two replays differ (2 % at that shape); short steps (QWS, Normal) keep running free.
        env = os.environ.get("GNNPN_PIPE_LOCKSTEP")
        long_steps = int(getattr(pipe.low.actor, "seq_len", 0)) >= 2000          # recurrent steps per problem (T * K)
        self.lockstep = self.n_slots == 2 and self.n_streams == 2 and (env == "1" or (env is None and long_steps))
        self._open_leader, self._last_done = None, [None] * max(2, self.n_slots)
        self._slot_done = [None] * self.n_slots
        self._deferred = None                    # (slot, after): a leader whose replay waits for its partner's submission (submit)
        self._copy_streams = [torch.cuda.Stream() for _ in range(self.n_slots)]   # host-to-device transfers of pinned arenas (submit)
        # Common start of a burst (two free-running slots only).  Two slots that begin a burst a host enqueue apart (~0.1 ms:
        # the second replay is not in its queue yet when the first starts) run in step for a few steps and then slip apart ONCE
        # — one slot's step takes 1.7 instead of 1.2 ms at the QWS shape — which costs a 20-step burst 0.5 ms (4 %); slots
        # released TOGETHER stay in step (tools/probes/stagger_probe.py: 11.50 against 12.01 ms per 20 steps, the hold included;
        # 30 and 60 us are too short, 90 and more work).  So the first replay after the runner has been idle is held behind a
        # gate on another stream, and so is the other slot's first one; later replays are not.  The gate (gnnpn_gate_wait: one
        # wavefront polling a word in pinned host memory) is opened by the HOST the moment the second replay is in its queue —
        # or by synchronize / poll, or after COMMON_START_US at the latest (a burst of one replay that is waited for some other way).
        # "Idle" = since the last synchronize() / poll() of this runner: a caller that works in bursts waits for them that way.
        self.common_start_us = float(os.environ.get("GNNPN_PIPE_COMMON_START_US", COMMON_START_US)) \
            if (self.n_slots == 2 and self.n_streams == 2 and not self.lockstep) else 0.0
        self._gate = None                        # [event, slots still to be held behind it]
        self._gate_flag = torch.zeros(1, dtype=torch.int32).pin_memory() if self.common_start_us > 0 else None   # the word the gate polls
        self._gate_seq = 0                       # value that opens the current gate (one more per burst)
        self._drained = True                     # nothing in flight: set by synchronize / poll, cleared by the next replay
        # (the gate's spin runs on slot 0's transfer stream, idle whenever the batches are resident: one more stream of its own
        # changed which streams share a hardware queue and cost the pinned-host path 1.7 %)
        self._gate_stream = self._copy_streams[0] if self.common_start_us > 0 else None
        # gnnpn_launch_opts_t.paired_start: half-batches (set inside ML2PNPipeline.run) and slots started in pairs begin together
        self.batches = [self._clone(example_batch) for _ in range(self.n_slots)]   # never alias caller tensors
        self.workspaces = [ops.new_workspaces(example_batch.x.device) for _ in range(2 if self.halves else self.n_slots)]
        # (every graph keeps its OWN memory pool, also the half-batch mode's two that never run together: in a shared pool
        # the second capture places its outputs where the first keeps intermediates, and the first graph's next replay
        # writes over them — measured: garbage in a slot's outputs one submission later)
        # write_through: the placement-independent hand-off form (agent-scope write-through granule stores) in every cooperative
        # launch of this runner — the degraded mode bench.py falls back to when a launch reported a failed hand-off
        self.write_through = DEFAULT_WRITE_THROUGH if write_through is None else bool(write_through)
        # auto_degrade: the first poll / synchronize that finds a failed launch (any status code, the shortfall of finished
        # workgroup-tiles included) re-captures every slot with the write-through hand-off; ``degraded`` keeps the status that
        # caused it.  The batches since the previous poll are invalid either way and the caller submits them again.
        self.auto_degrade, self.degraded = bool(auto_degrade), None
        self._capture_graphs()

    def _capture_graphs(self):
        with ops.lds_footprint(self.front_lds_kb):            # captured launches keep the footprint (a per-launch attribute)
            self.graphs = [self.pipe.capture(self.services, self.batches[s], decode_impl=self.decode_impl, lds_kb=self.lds_kb[s],
                                             ws=tuple(self.workspaces) if self.halves else self.workspaces[s], paired_start=self.lockstep,
                                             write_through=self.write_through)
                           for s in range(self.n_slots)]

    def _degrade(self, word):
        """A launch of this runner failed (status ``word``): from now on every cooperative launch uses the placement-independent
        write-through hand-off (new graphs over the same static inputs and workspaces; the old ones are dropped)."""
        if not self.auto_degrade or self.write_through:
            return
        import warnings
        warnings.warn(f"gnnpn: PipelinedRunner: cooperative launches reported status {word:#x}; switching this runner to the "
                      f"write-through hand-off (the batches since the last poll / synchronize must be submitted again)", RuntimeWarning)
        self.write_through, self.degraded = True, word
        self._open_leader, self._last_done = None, [None] * max(2, self.n_slots)
        self._slot_done = [None] * self.n_slots
        self._capture_graphs()

    @staticmethod
    def _fields(b):
        return (b.x, b.wf_csr.rowptr, b.wf_csr.col, b.seg_ptr, b.local_bounds, b.present, b.global_bounds)

    @classmethod
    def _clone(cls, b, host=False):
        """A copy of the batch whose seven tensors are views into ONE allocation (``_arena``, 256-byte aligned pieces): a
        batch that was packed the same way (``pack``) moves into a slot's static inputs with one copy instead of seven —
        at a 0.55 ms step seven 5 us copy kernels in front of every replay are 3 % of the slot's cycle.  ``host``: the
        arena in PINNED host memory (``pack_host``): the same single copy, now host-to-device."""
        fields = cls._fields(b)
        offs, total = [], 0
        for t in fields:
            offs.append(total)
            total += (t.numel() * t.element_size() + 255) // 256 * 256
        arena = (torch.empty(max(total, 256), dtype=torch.uint8, pin_memory=True) if host else
                 torch.empty(max(total, 256), dtype=torch.uint8, device=b.x.device))
        views = []
        for t, o in zip(fields, offs):
            v = arena[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape)
            v.copy_(t)
            views.append(v)
        x, rowptr, col, seg, lb, pr, gb = views
        out = DeviceBatch(x, graph.CSR(rowptr, col, None, b.wf_csr.n), seg, lb, pr, gb, b.max_nodes)
        out._arena, out._layout = arena, tuple((tuple(t.shape), t.dtype) for t in fields)
        return out

    def pack(self, batch):
        """The batch in the slots' own memory layout: ``submit`` moves such a batch with a single device-to-device copy."""
        return self._clone(batch)

    def pack_host(self, batch):
        """The batch (host or device tensors) in the slots' layout in ONE pinned host allocation: ``submit`` moves it into a
        slot's static inputs with a single asynchronous host-to-device copy on the slot's stream — the reference's per-batch
        ``inputs.cuda()`` (src/models/trainPNHigh.py:134-136) as one copy instead of seven.  The caller must not rewrite the
        arena before that copy has run (record an event on ``stream(slot)``)."""
        return self._clone(batch, host=True)

    def submit(self, batch=None, after=None):
        """Enqueue one batch; returns (outputs dict, slot).  The outputs are the slot's static tensors:
        consume them (or record an event on ``stream(slot)``) before the slot comes round again, ``n_slots`` submits
        later — in every mode (half-batch mode included: there the slots alternate on one stream).
        ``after(outputs, slot)``: called with the slot's stream current, right behind the enqueued replay — the place for the
        device-to-host copies of the results.  It matters for slots that start in PAIRS (``lockstep``) when the batch comes from a
        pinned host arena: the leader's replay is then enqueued together with its partner's, once BOTH transfers are on their way
        (each replay waits for both), so that the pair still starts together; ``after`` of the leader runs at that moment.
        Touching ``stream(slot)``, ``poll`` or ``synchronize`` enqueues a waiting leader at once (alone)."""
        s = self.count % self.n_slots
        if self.count == 0 and self._priority_chosen and self.stream_priority == 0 and self.n_streams > 1 and _has_collective_stream() \
                and int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8:
            import warnings                                # the process group came AFTER this runner: its streams were created at normal priority
            warnings.warn("PipelinedRunner was created before the RCCL process group: its slots' streams share hardware queues with the "
                          "collective's stream (about 6 % slower with one all-gather per 8 steps at the QWS shape); create the runner after "
                          "init_process_group, or pass stream_priority=-1", RuntimeWarning, stacklevel=2)
        self._take_turn()
        self.count += 1
        if self._deferred is not None and self._deferred[0] == s:
            self._flush_deferred()
        staged = False                                   # a transfer of this batch is on the slot's copy stream
        st = self._stream(s)
        with torch.cuda.stream(st):
            if batch is not None:
                dst = self.batches[s]
                lim = ops.REQUEST_BRANCH_MAX_NODES        # the captured graph holds the one-launch GIN branch (small
                if 0 < dst.max_nodes <= lim and not 0 < batch.max_nodes <= lim:   # workflow graphs) or the layered kernels
                    raise ops.GnnpnError(f"PipelinedRunner: the captured graph holds the one-launch GIN branch (graphs of <= "
                                         f"{lim} nodes); this batch has max_nodes = {batch.max_nodes}")
                if getattr(batch, "_layout", None) is not None and batch._layout == dst._layout and \
                        (batch._arena.device == dst._arena.device or (batch._arena.device.type == "cpu" and batch._arena.is_pinned())):
                    if batch._arena.device.type == "cpu" and HOST_COPY_ON_ITS_OWN_STREAM:
                        # pinned host arena: the copy engine's transfer goes on a stream of its own, behind the replay that last read
                        # this slot's static inputs and in front of its next one (events) — in half-batch mode the next batch then
                        # crosses PCIe under the current batch's kernels
                        cs = self._copy_streams[s]
                        if self._slot_done[s] is not None:
                            cs.wait_event(self._slot_done[s])      # the replay that last read this slot's static inputs
                        else:
                            cs.wait_stream(st)
                        with torch.cuda.stream(cs):
                            dst._arena.copy_(batch._arena, non_blocking=True)
                        st.wait_stream(cs)
                        staged = True
                    else:
                        dst._arena.copy_(batch._arena, non_blocking=True)      # packed alike: one copy (device-to-device, or pinned host to device)
                else:
                    for a, b in zip(self._fields(dst), self._fields(batch)):
                        if a.shape != b.shape:
                            raise ops.GnnpnError(f"PipelinedRunner: batch shape {tuple(b.shape)} != captured {tuple(a.shape)}")
                        a.copy_(b, non_blocking=True)
        if self.lockstep and staged:
            # Pairs and transfers: two transfers of a pair run one after the other on the copy engine (0.7 ms each at the 2000-task
            # shape), so a leader that starts behind ITS transfer is that far ahead of its partner, its encoder arrives while the
            # partner's front half still holds CUs, and the common start the pairing exists for is gone (PCIe-inclusive rate 0.57
            # of the resident one).  The leader's replay therefore waits here, on the host side, for its partner's submission.
            if self._deferred is None:
                self._deferred = (s, after)
                return self.graphs[s].outputs, s
            lead, after_lead = self._deferred
            self._deferred = None
            return self._replay_pair(lead, after_lead, s, after), s
        if self._deferred is not None:
            self._flush_deferred()
        if batch is not None and self.common_start_us > 0:
            arena = getattr(batch, "_arena", None)
            if (arena if arena is not None else batch.x).device.type == "cpu":
                self._drained = False                # batches from host memory: no common start (their transfers share the copy
                #                                      engine, slots in step wait for each other's: 436 against 442 k problems/s)
        return self._replay(s, after), s

    def _replay(self, s, after=None):
        """One slot's replay, with the pairing of ``lockstep`` for submissions that come one at a time."""
        st = self._stream(s)
        with torch.cuda.stream(st):
            leader = False
            if self.lockstep:
                # A submission joins the leader that is waiting for a partner (and starts with it) if that leader has not
                # finished yet — the host runs ahead of the device, so back-to-back submissions always pair —; otherwise it
                # leads a new pair, behind whatever the other slot ran last.
                lead = self._open_leader
                if lead is not None and lead[0] != s and not lead[2].query():
                    st.wait_event(lead[1])
                    self._open_leader = None
                else:
                    leader = True
                    if self._last_done[1 - s] is not None:
                        st.wait_event(self._last_done[1 - s])
                    started = torch.cuda.Event()
                    started.record(st)
            last_held = self.common_start_us > 0 and self._hold_for_common_start(s, st)
            out = self.graphs[s]()
            if last_held:
                self._open_gate()                   # both held replays are in their queues: they start now, together
            done = torch.cuda.Event()
            done.record(st)
            self._slot_done[s] = done               # what the next transfer into this slot's static inputs waits for
            if self.lockstep:
                self._last_done[s] = done
                if leader:
                    self._open_leader = (s, started, done)
            if after is not None:
                after(out, s)
        return out

    def _hold_for_common_start(self, s, st):
        """First replay after the runner was waited for: a gate (see __init__) in front of this replay and the other slot's next
        one.  Returns True if this replay completes the pair — the caller opens the gate once the replay is enqueued."""
        if self._gate is None and self._drained:
            # (Deliberately NOT "whenever the streams happen to be empty": a host-paced caller whose device catches up now and
            # then would pay the hold every time — measured with the pinned-host batches of tools/bench_pcie.py: -2.7 %.)
            self._gate_seq = (self._gate_seq + 1) & 0x7fffffff
            ev = torch.cuda.Event()
            with torch.cuda.stream(self._gate_stream):
                _lib.check(_lib.load().gnnpn_gate_wait(ctypes.c_void_p(self._gate_flag.data_ptr()), self._gate_seq, int(self.common_start_us),
                                                       _lib.stream_ptr()), "gnnpn_gate_wait")
                ev.record(self._gate_stream)
            self._gate = [ev, set(range(self.n_slots))]
        self._drained = False
        if self._gate is not None and s in self._gate[1]:
            st.wait_event(self._gate[0])
            self._gate[1].discard(s)
            return not self._gate[1]
        return False

    def _open_gate(self):
        """The host opens the pending gate (both replays are enqueued, or the caller is about to wait)."""
        if self._gate is not None:
            self._gate_flag[0] = self._gate_seq
            self._gate = None

    def _replay_pair(self, lead, after_lead, s, after):
        """Leader and partner enqueued together: each replay behind BOTH transfers, the leader behind whatever the partner's slot
        ran last, the partner behind the leader's start."""
        st_l, st_f = self._stream(lead), self._stream(s)
        st_l.wait_stream(self._copy_streams[s])
        st_f.wait_stream(self._copy_streams[lead])
        if self._last_done[s] is not None:
            st_l.wait_event(self._last_done[s])
        started = torch.cuda.Event()
        started.record(st_l)
        st_f.wait_event(started)
        outs = {}
        for slot, stream, cb in ((lead, st_l, after_lead), (s, st_f, after)):
            with torch.cuda.stream(stream):
                outs[slot] = self.graphs[slot]()
                done = torch.cuda.Event()
                done.record(stream)
                self._slot_done[slot] = self._last_done[slot] = done
                if cb is not None:
                    cb(outs[slot], slot)
        self._open_leader = None
        return outs[s]

    def _flush_deferred(self):
        if self._deferred is not None:
            lead, after_lead = self._deferred
            self._deferred = None
            self._replay(lead, after_lead)

    def _take_turn(self):
        """Two runners of one process take TURNS on a device.  A CU holds two cooperative workgroups, which is what ONE runner's two
        launches in flight use; two runners fed alternately (two models served from one process) put four launches in front of those
        two slots, launches stay half-staffed behind each other and the bounded waits give up — loud, and very slow (round 6,
        tools/probes/dbg_two_runners.py: 200 alternating steps 3.6 s with status 0x13 against 0.11 s one runner after the other).
        So the first submit after ANOTHER runner of the device was the last to submit waits (stream-side: events, no host
        synchronisation) for everything that runner has enqueued.  Switching runners therefore drains the pipeline — a caller who
        alternates per step runs one step at a time — but every launch finds the slots it was built for."""
        import weakref
        dev = self.batches[0].x.device
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        ref = _last_runner_on_device.get(key)
        other = ref() if ref is not None else None
        if other is not self:
            if other is not None:
                other._flush_deferred()
                other._open_gate()
                for ost in other.streams:
                    ev = torch.cuda.Event()
                    ev.record(ost)
                    for st in self.streams:
                        st.wait_event(ev)
            _last_runner_on_device[key] = weakref.ref(self)

    def _stream(self, slot):
        return self.streams[slot % self.n_streams]

    def stream(self, slot):
        """The HIP stream slot ``slot``'s replays run on (half-batch mode: every slot's, there is one step in flight).  A leader
        that is waiting for its partner (``submit``) is enqueued first: work a caller puts on the stream comes behind the replay."""
        self._flush_deferred()
        return self._stream(slot)                  # (a pending common-start gate stays: it opens with the partner's replay, or by its time-out)

    def reference_run(self, slot=0):
        """The same kernels on ONE stream, nothing overlapped (used to check an overlapped result)."""
        return self.pipe.run(self.services, self.batches[slot], decode_impl=self.decode_impl, write_through=self.write_through)   # one launch per kernel, whole batch

    def poll(self):
        """Wait for every slot's stream; the OR of the slots' sticky status words since the last poll / check, cleared — 0: no
        launch reported a failed hand-off.  The non-raising form of ``synchronize(check=True)``."""
        self._flush_deferred()
        self._open_gate()
        for st in self.streams:
            st.synchronize()
        self._drained = True
        word = 0
        for w in self.workspaces:
            word |= w.poll()
        if word:
            self._degrade(word)
        return word

    def progress(self):
        """The proof-of-work counters the last poll / synchronize read, per workspace (ops.Workspaces.last_progress)."""
        return [w.last_progress for w in self.workspaces]

    def synchronize(self, check=True):
        """Wait for every slot's stream; with ``check`` raise if any launch of any slot since the last call reported a
        failed inter-workgroup hand-off (its outputs would be garbage) — the sticky status words of the slots."""
        self._flush_deferred()
        self._open_gate()
        for st in self.streams:
            st.synchronize()
        self._drained = True
        if check:
            err = None
            for w in self.workspaces:
                try:
                    w.check("PipelinedRunner")
                except ops.GnnpnError as e:      # every workspace is read (and cleared) before the first failure is raised
                    err = err or e
            if err is not None:
                self._degrade(int(getattr(err, "status", 0xffff)))
                raise err
