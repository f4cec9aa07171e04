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
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class GradSync:
    """Asynchronous bucketed all-reduce (sum) of a flat gradient buffer."""

    def __init__(self, flat_grads: torch.Tensor, bucket_bytes: int = 48 << 20, group=None):
        self.flat = flat_grads
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = flat_grads.numel()
        per = max(1, bucket_bytes // flat_grads.element_size())
        self.bounds: List[tuple] = [(i, min(n, i + per)) for i in range(0, n, per)]
        self._works = []

    def launch(self) -> None:
        """Enqueue the reduction behind everything already queued on the current stream; returns immediately."""
        self._works = []
        if not dist.is_initialized():
            return  # single process without a group; a 1-rank group still runs the (trivial) collective
        for a, b in self.bounds:
            self._works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self) -> None:
        """Make the current stream (not the host) wait for the reduction."""
        for w in self._works:
            w.wait()
        self._works = []

    @property
    def scale(self) -> float:
        """all_reduce sums; the optimizer kernel folds the 1/world mean in (no extra pass over the gradients)."""
        return 1.0 / self.world


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
