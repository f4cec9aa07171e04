"""Data parallelism for the distillation step: one process per GPU, RCCL over xGMI.

The reference gets data parallelism implicitly (Lightning ``strategy='ddp'``, train.py:494 -> torch DDP -> NCCL):
one all-reduce(mean) of the student gradients per optimizer step and a parameter broadcast at construction.
Here the gradients already live in ONE flat fp32 buffer, so the exchange is a handful of large all-reduces
(``bucket_bytes`` each; xGMI is point-to-point, a ring is per-link bound, so few big messages beat DDP's 25 MB
default) issued asynchronously right after the backward; the caller then enqueues the NEXT batch's frozen-teacher
forward, and only waits for the reduction just before the optimizer kernel -- so the collective is hidden
behind teacher compute.  ``backend='nccl'`` is RCCL on ROCm; tests run the same code on ``gloo``/CPU tensors.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple:
    """(rank, world, local_rank) from the torchrun environment; initialises the default group when WORLD_SIZE > 1."""
    # the host driver of this pool only supports dmabuf IPC; RCCL's peer mappings fail with the legacy mode (set before HIP starts)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        # HIP streams are multiplexed onto GPU_MAX_HW_QUEUES hardware queues (default 4): the step already uses three (student,
        # teacher, parameter-gradient side stream) and RCCL adds its own -- streams that share a queue serialise
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:   # FH_DIST_BACKEND=gloo: CPU-staged collectives on device tensors (two ranks on ONE GPU, tests)
            backend = os.environ.get("FH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class GradSync:
    """Asynchronous bucketed all-reduce (sum) of a flat gradient buffer.

    Two ways to start it: ``launch()`` -- the whole buffer, behind everything queued on the current stream; or, from inside the
    native backward (``fh_student_backward_notify``), ``begin()`` then one ``on_segment(seg, first, count)`` per gradient range as
    it becomes final (projection heads first, transformer layers next, encoder front last): the library has already made
    ``comm_stream`` wait for the kernels that write the range, and the collective is enqueued under that stream -- so the heads'
    and the layers' buckets are on the wire while the conv frontend is still in backward.  ``wait()`` makes the current stream
    (not the host) wait for every bucket."""

    def __init__(self, flat_grads: torch.Tensor, bucket_bytes: int = 48 << 20, group=None, wire_dtype: Optional[torch.dtype] = None):
        self.flat = flat_grads
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # Optional 16-bit exchange (``wire_dtype=torch.bfloat16``; FH_DP_GRAD_WIRE=bf16 in W2V2Distil): each range is rounded into a
        # bf16 staging buffer on the communication stream, all-reduced there (62 MB instead of 126 MB per step over xGMI) and widened
        # back into the fp32 buffer in front of AdamW, whose moments and master weights stay fp32.  Every rank receives the same
        # reduced values, so replicas stay bit-identical; the gradient itself carries bf16's 8 significant bits per rank term (what
        # torch DDP's bf16_compress_hook does).  Off by default: the fp32 exchange is the reference's (train.py:494).
        self.wire = torch.empty(flat_grads.numel(), dtype=wire_dtype, device=flat_grads.device) if wire_dtype is not None else None
        self._wire_ranges: List[tuple] = []
        self.per = max(1, bucket_bytes // (self.wire if self.wire is not None else flat_grads).element_size())
        n = flat_grads.numel()
        self.bounds: List[tuple] = [(i, min(n, i + self.per)) for i in range(0, n, self.per)]
        self._works = []
        self.launched: List[tuple] = []   # (first, last) element ranges of the current exchange, in launch order
        self.comm_stream = None
        if flat_grads.is_cuda and dist.is_initialized():
            self.comm_stream = torch.cuda.Stream(device=flat_grads.device)

    @property
    def active(self) -> bool:
        return dist.is_initialized()

    @property
    def comm_handle(self) -> Optional[int]:
        return self.comm_stream.cuda_stream if self.comm_stream is not None else None

    def begin(self) -> None:
        self._works = []
        self.launched = []
        self._wire_ranges = []

    def on_segment(self, segment: int, first: int, count: int) -> None:
        """All-reduce flat[first : first + count] in buckets; ordered behind ``comm_stream`` when there is one."""
        if not dist.is_initialized():
            self.launched.append((first, first + count))
            return
        ctx = torch.cuda.stream(self.comm_stream) if self.comm_stream is not None else _NullCtx()
        with ctx:
            for a in range(first, first + count, self.per):
                b = min(first + count, a + self.per)
                if self.wire is not None:
                    self.wire[a:b].copy_(self.flat[a:b])          # fp32 -> bf16 rounding, on the communication stream
                    self._works.append(dist.all_reduce(self.wire[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    self._wire_ranges.append((a, b))
                else:
                    self._works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                self.launched.append((a, b))

    def launch(self) -> None:
        """Enqueue the reduction of the whole buffer behind everything already queued on the current stream; returns at once."""
        self.begin()
        if not dist.is_initialized():
            return  # single process without a group; a 1-rank group still runs the (trivial) collective
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.flat.device))
        self.on_segment(0, 0, self.flat.numel())

    def wait(self) -> None:
        """Make the current stream (not the host) wait for the reduction."""
        for w in self._works:
            w.wait()
        self._works = []
        for a, b in self._wire_ranges:                           # widen the reduced 16-bit sums back (behind the waits, same stream)
            self.flat[a:b].copy_(self.wire[a:b])
        self._wire_ranges = []

    def covered(self) -> bool:
        """True when the ranges launched since ``begin()`` tile the whole buffer exactly once."""
        pos = 0
        for a, b in sorted(self.launched):
            if a != pos:
                return False
            pos = b
        return pos == self.flat.numel()

    @property
    def scale(self) -> float:
        """all_reduce sums; the optimizer kernel folds the 1/world mean in (no extra pass over the gradients)."""
        return 1.0 / self.world


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, group=None) -> None:
    """DDP's construction-time broadcast: rank 0's weights win."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)


def assert_replicas_equal(flat_params: torch.Tensor, group=None) -> None:
    """Cheap drift check: max |p - p_rank0| must be 0 (data-parallel replicas apply identical updates)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    ref = flat_params.clone()
    dist.broadcast(ref, src=0, group=group)
    diff = (ref - flat_params).abs().max()
    dist.all_reduce(diff, op=dist.ReduceOp.MAX, group=group)
    if diff.item() != 0:
        raise RuntimeError(f"data-parallel replicas diverged: max |delta| = {diff.item()}")
