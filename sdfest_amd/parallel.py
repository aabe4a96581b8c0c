"""Multi-GPU layer of the hot path: views shard, one gradient is shared.

The reference is single-process single-GPU and loops over views in Python
(estimation/simple_setup.py:420-446).  Here each rank (one process per GPU) renders a
contiguous shard of the views; forward needs no communication at all, and backward has
exactly one exchange: the d/dSDF volume is summed over ranks with ONE all-reduce (RCCL over
xGMI on GPUs: torch.distributed backend "nccl"; gloo in the tests).  1 MiB is latency-bound on
7x153 GB/s links, so whatever else must be shared rides in the same flat bucket: in the sharded
render-and-compare loop (``pipeline.FusedRenderAndCompare(process_group=...)``) one 20-float record per
view -- its pose sums and loss values, non-zero on the rank that owns the view -- behind the volume
(``allreduce_bucket``; as 64-bit integers in the deterministic mode, where the sum is then exact and the
whole trajectory bitwise independent of the split).  Parameters and optimiser state are replicated and
never broadcast.
"""
from typing import Optional, Sequence, Tuple

import torch


def shard_views(n_views: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[begin, end) of the contiguous view shard owned by `rank` (remainder spread left)."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, rem = divmod(n_views, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def resolve_group(process_group):
    """(group, rank, world_size) of a caller's ``process_group`` argument: None -> single process (no exchange at
    all); "world" -> the default group of an initialised torch.distributed; otherwise a ProcessGroup."""
    if process_group is None:
        return None, 0, 1
    dist = _dist()
    if dist is None:
        raise RuntimeError("a process group was given but torch.distributed is not initialised "
                           "(call torch.distributed.init_process_group first: backend 'nccl' is RCCL on ROCm)")
    group = dist.group.WORLD if isinstance(process_group, str) and process_group == "world" else process_group
    return group, dist.get_rank(group), dist.get_world_size(group)


def allreduce_bucket(bucket: torch.Tensor, group=None, integer: bool = False) -> torch.Tensor:
    """The sharded loop's ONE exchange per iteration: sum `bucket` over the ranks of `group`, in place.

    integer=True: the bucket is summed as 64-bit integers (its bytes reinterpreted) -- exact and order-independent
    for the deterministic mode's fixed-point volume, and for float words that are non-zero on one rank only (the view
    records): x + 0 + ... + 0 reproduces x bit for bit whatever it holds."""
    dist = _dist()
    if dist is None or group is None:
        return bucket
    flat = bucket.view(-1)
    if integer:
        if flat.dtype != torch.int64:
            if (flat.numel() * flat.element_size()) % 8:
                raise RuntimeError("an integer exchange needs a bucket of a multiple of 8 bytes")
            flat = flat.view(torch.int64)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return bucket


class _Done:
    """Handle of an exchange that needed no communication."""

    def wait(self):
        return True


def allreduce_shared_gradients(g_sdf: torch.Tensor,
                               extra: Optional[Sequence[torch.Tensor]] = None, group=None,
                               async_op: bool = False):
    """Sum the shared gradients over ranks, in place.  No-op when not distributed.

    g_sdf: this rank's d/dSDF (sum over its own views).  extra: small tensors shared by all
    ranks (e.g. the world-frame pose gradient of a single object seen by many cameras); they
    ride in the same bucket so the step has exactly one collective.

    async_op=True (only without `extra`): returns a handle at once; ``handle.wait()`` makes the
    current stream wait for the sum.  The exchange then runs beside whatever is launched in
    between -- e.g. the next step's forward, which does not read g_sdf (the backward, which
    overwrites it, must come after the wait).
    """
    dist = _dist()
    if async_op and extra:
        raise ValueError("async_op is only supported without extra tensors")
    if dist is None:
        return _Done() if async_op else g_sdf
    if not extra:
        work = dist.all_reduce(g_sdf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return work if async_op else g_sdf
    flat = torch.cat([g_sdf.reshape(-1)] + [e.reshape(-1) for e in extra])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    n = g_sdf.numel()
    g_sdf.copy_(flat[:n].view_as(g_sdf))
    for e in extra:
        e.copy_(flat[n:n + e.numel()].view_as(e))
        n += e.numel()
    return g_sdf


def allreduce_fixed_gradients(fixed: torch.Tensor, out: torch.Tensor, group=None) -> torch.Tensor:
    """Deterministic mode (``SDF_GRAD_DETERMINISTIC``): sum the ranks' int64 fixed-point d/dSDF volumes
    (``BatchRenderPlan.g_sdf_fixed()``) with ONE all-reduce and convert the sum to float into ``out``
    (``sdfr_fixed_to_float``).  Integer addition is associative, so the result is bitwise the same for every
    split of the views over ranks -- and the same as a single process rendering all of them."""
    from . import _lib
    dist = _dist()
    if fixed.dtype != torch.int64 or not fixed.is_contiguous() or not fixed.is_cuda:
        raise RuntimeError("fixed must be a contiguous int64 CUDA tensor")
    if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != fixed.numel() or out.device != fixed.device:
        raise RuntimeError("out must be a contiguous float32 tensor of the same size on the same device")
    if dist is not None:
        dist.all_reduce(fixed, op=dist.ReduceOp.SUM, group=group)
    rc = _lib.lib().sdfr_fixed_to_float(fixed.data_ptr(), fixed.numel(), out.data_ptr(), fixed.device.index,
                                        torch.cuda.current_stream(fixed.device).cuda_stream)
    _lib.check(rc, "sdfr_fixed_to_float")
    return out


def spawn_ranks(cmd: Sequence[str], world_size: int, master_port: Optional[int] = None,
                env: Optional[dict] = None, timeout: Optional[float] = None) -> int:
    """Start `world_size` fresh processes of `cmd`, one per GPU, with the torch.distributed
    environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT) set, wait for
    all of them and return 0 only if every one exited 0.

    The caller must not have touched the GPU: the workers are plain child processes
    (``subprocess.Popen``, no fork of an initialised runtime, no exec of a running one), the
    same thing ``python -m torch.distributed.run --nproc-per-node N`` starts.  Their stdout and
    stderr are inherited, so rank 0's result line reaches the caller's stdout unchanged.  When
    one worker fails, the others are terminated (exact PIDs) instead of waiting in a collective.
    """
    import os
    import socket
    import subprocess
    import time

    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    if master_port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            master_port = s.getsockname()[1]
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(world_size), LOCAL_WORLD_SIZE=str(world_size), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=str(master_port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(world_size):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(list(cmd), env=e))
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
        if rc != 0 or (deadline is not None and time.monotonic() > deadline):
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            return rc or 124
        if live:
            time.sleep(0.02)
    return rc
