"""Clip-sharded execution across the GPUs of one node.

The reference has no distributed code at all.  Clips are
independent on this path (reference stft.py:99 flattens the batch; dgt.py:137-141
loops over clips), so a batch shards by contiguous blocks of clips: one process
per GPU, no data-path collective.  The only exchange is an optional RCCL
all-gather that reassembles the per-rank feature shards on every rank
(`torch.distributed` backend "nccl" == RCCL over xGMI; "gloo" in the CPU tests).
"""
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_clips: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous split [lo, hi) of the flattened batch axis; the first n % world ranks get one extra clip."""
    base, extra = divmod(n_clips, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(x: torch.Tensor, rank: Optional[int] = None, world_size: Optional[int] = None,
                group=None) -> torch.Tensor:
    """This rank's clips of a (B, ...) batch (a view, no copy).  rank / world_size default to those of `group`
    (the default group when None)."""
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_bounds(x.shape[0], rank, world_size)
    return x[lo:hi]


class _RaggedGather:
    """Work handle of the ragged all-gather: `wait()` waits for the collective, then compacts the padded
    per-rank blocks into `out` (rows [lo_r, hi_r) of every rank) on the current stream."""

    def __init__(self, work, buf, out, sizes, mx):
        self.work, self.buf, self.out, self.sizes, self.mx = work, buf, out, sizes, mx

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
            lo = 0
            for r, n in enumerate(self.sizes):
                self.out[lo:lo + n].copy_(self.buf[r * self.mx:r * self.mx + n])
                lo += n
            self.buf = None
        return True


def all_gather_features(local: torch.Tensor, n_clips_total: int, group=None, async_op: bool = False):
    """Reassemble (B_total, ...) features from per-rank shards on every rank.

    Equal shards use one `all_gather_into_tensor` (a single RCCL collective writing
    straight into the output); ragged shards are padded to the largest shard first and
    compacted after the collective.  Returns the gathered tensor -- and, when async_op, a
    handle whose `wait()` must be called before the tensor is read (on the ragged path the
    compaction runs inside that `wait()`)."""
    if not dist.is_initialized():
        return (local, None) if async_op else local
    world = dist.get_world_size(group)          # a one-rank group still runs the collective (a copy): same code path
    rank = dist.get_rank(group)
    sizes = [shard_bounds(n_clips_total, r, world)[1] - shard_bounds(n_clips_total, r, world)[0] for r in range(world)]
    assert local.shape[0] == sizes[rank], "local shard has %d clips, expected %d" % (local.shape[0], sizes[rank])
    local = local.contiguous()
    if len(set(sizes)) == 1:
        out = torch.empty((n_clips_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        work = dist.all_gather_into_tensor(out, local, group=group, async_op=async_op)
        return (out, work) if async_op else out
    mx = max(sizes)
    padded = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:sizes[rank]] = local
    buf = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    out = torch.empty((n_clips_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    handle = _RaggedGather(dist.all_gather_into_tensor(buf, padded, group=group, async_op=True), buf, out, sizes, mx)
    if async_op:
        return out, handle
    handle.wait()
    return out


def sharded_apply(fn: Callable[[torch.Tensor], torch.Tensor], x: torch.Tensor, gather: bool = True, group=None):
    """Run `fn` (e.g. a ComposeAudioTransform) on this rank's clips of the replicated batch x;
    gather=True returns the full result on every rank, gather=False leaves it sharded."""
    local = fn(shard_batch(x, group=group))
    return all_gather_features(local, x.shape[0], group=group) if gather else local
