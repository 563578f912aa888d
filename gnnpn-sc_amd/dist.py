"""One process per GPU.  Composition problems are independent given the (replicated) weights and
service table, so a batch shards contiguously across ranks with NO collective inside the path;
the only exchange is ONE all-gather of the selected indices [B/N, T] int32 at the end
(RCCL over xGMI on the GPU box — backend "nccl" is RCCL on ROCm; gloo in the CPU tests).
The reference has no multi-device code; this module is new.
"""
import os

import torch
import torch.distributed as td


def init_process_group(backend, device=None):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:    # a lone process asking for a group: world of one
        import socket
        os.environ.update(RANK="0", WORLD_SIZE="1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    if not td.is_initialized():
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        td.init_process_group(backend=backend, **kw)
    return td.get_rank(), td.get_world_size()


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n problems for ``rank`` (same split as DeviceBatch.shard)."""
    return n * rank // world, n * (rank + 1) // world


def all_gather_indices(idx_local, sizes=None):
    """idx_local [b_r, T] int32 -> [sum b_r, T] on every rank, rank order = problem order.
    Equal shards use one all_gather_into_tensor; ragged shards pad to the largest."""
    world = td.get_world_size()
    if idx_local.is_cuda and td.get_backend() == "gloo":      # launch-path checks without RCCL: stage through the host
        return all_gather_indices(idx_local.cpu(), sizes).to(idx_local.device)
    if sizes is None or len(set(sizes)) == 1:
        out = torch.empty((world * idx_local.shape[0],) + tuple(idx_local.shape[1:]), dtype=idx_local.dtype,
                          device=idx_local.device)
        td.all_gather_into_tensor(out, idx_local.contiguous())
        return out
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(idx_local.shape[1:]), dtype=idx_local.dtype, device=idx_local.device)
    pad[: idx_local.shape[0]] = idx_local
    out = torch.empty((world * m,) + tuple(idx_local.shape[1:]), dtype=idx_local.dtype, device=idx_local.device)
    td.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * m: r * m + sizes[r]] for r in range(world)])


def all_gather_indices_async(idx_local, out=None):
    """Equal-shard all-gather that does NOT make the calling stream wait for the collective: returns (out, work).  torch's
    RCCL process group runs collectives on its own stream, ordered after the work already queued on the calling stream;
    with ``async_op`` the caller's stream is free to go on (the next pipelined step) while the 48 KB gather crosses xGMI.
    Call ``work.wait()`` (a stream-side wait, not a host block) on whatever stream next reads ``out`` or overwrites
    ``idx_local``.  Backends without device collectives (gloo launch-path checks) fall back to the blocking form."""
    if idx_local.is_cuda and td.get_backend() == "gloo":
        return all_gather_indices(idx_local), None
    world = td.get_world_size()
    if out is None:
        out = torch.empty((world * idx_local.shape[0],) + tuple(idx_local.shape[1:]), dtype=idx_local.dtype,
                          device=idx_local.device)
    work = td.all_gather_into_tensor(out, idx_local.contiguous(), async_op=True)
    return out, work


def barrier(world):
    if world > 1:
        td.barrier()


def max_over_ranks(value, device, world):
    if world <= 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if td.get_backend() == "gloo" else device)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def destroy(world):
    if world > 1 and td.is_initialized():
        td.destroy_process_group()
