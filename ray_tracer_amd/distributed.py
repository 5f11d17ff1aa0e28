"""Multi-GPU sharding of one image: progressive frames are split across ranks and the
float accumulation images are summed with one reduce.

The shader seeds every pixel with ``pixelIndex + frame * 719393`` (ray_tracing.comp:80-82)
and averages frames with equal weight (:106-107), so frames are independent sample
batches: rank r of W renders frames r, r+W, r+2W, ... into a private accumulation image
and one sum over ranks gives the same image one device would accumulate (up to the
order of float additions).  Splitting inside a frame's sample loop would not reproduce
the RNG streams, because a pixel's state advances sequentially through its samples.

One process per GPU; the reduce is ``torch.distributed.reduce`` (RCCL over xGMI with the
``nccl`` backend, gloo on CPU in the tests).  start_reduce / finish_reduce split it so
that a renderer with two accumulation images can trace the next batch while the previous
one is being summed.
"""
from __future__ import annotations

from typing import Tuple


def frame_partition(first_frame: int, n_frames: int, rank: int, world: int) -> Tuple[int, int, int]:
    """(first, count, stride) of the frames rank ``rank`` renders out of
    first_frame .. first_frame + n_frames - 1.  Ranks past the frame count get count 0."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"rank {rank} of world {world}")
    if n_frames < 0:
        raise ValueError("n_frames must be >= 0")
    count = (n_frames - rank + world - 1) // world if n_frames > rank else 0
    return first_frame + rank, count, world


def reduce_accumulation(accum, dst: int = 0):
    """Sum the accumulation images (any float tensor of identical shape on every rank)
    onto rank ``dst``.  Returns the tensor; its contents are the sum only on ``dst``."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def start_reduce(accum, dst: int = 0):
    """Begin the same sum without waiting for it: returns a handle for finish_reduce, or None
    when there is nothing to do (one rank).  With the nccl backend the sum runs on RCCL's own
    stream after the work already queued on the current stream, so kernels launched next --
    the following frame batch, into ANOTHER accumulation image -- overlap with it.  The
    tensor must not be touched until finish_reduce."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, async_op=True)
    return None


def finish_reduce(handle) -> None:
    """Order everything launched afterwards on the current stream behind the sum (nccl), or
    block until it is done (gloo)."""
    if handle is not None:
        handle.wait()


def render_split(renderer, accum, first_frame: int, n_frames: int, rank: int, world: int,
                 stage_on_host: bool = False) -> None:
    """One image over ``world`` ranks: this rank traces its share of the frames into ``accum`` (the
    torch tensor the renderer was created on, ``accum_buffer=accum.data_ptr()``), the images are
    summed onto rank 0, and rank 0 resolves the sum to its 8-bit image.  ``stage_on_host`` sums host
    copies instead (gloo, for ranks that share one device: a rehearsal of the control flow, RCCL
    needs one device per rank)."""
    import torch
    import torch.distributed as dist

    first, count, stride = frame_partition(first_frame, n_frames, rank, world)
    renderer.render(first, count, stride)
    if world > 1:
        if stage_on_host:
            renderer.sync()
            host = accum.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                accum.copy_(host)
                torch.cuda.synchronize(accum.device)
        else:
            finish_reduce(start_reduce(accum, dst=0))
    if rank == 0:
        renderer.resolve()
    renderer.sync()
