"""Time-slice sharding of a long recording across the GPUs of one node.

Lines of a spectrogram are independent (controllers/MainController.java:982-993
computes every ``waterfall[t]`` from its own sample span), so rank r of R takes
the contiguous line range ``[r*L//R, (r+1)*L//R)``; its input span overlaps the
next rank's by ``nfft - hop`` samples (a halo each rank reads or generates for
itself -- no input exchange).  The only collective on the spectrogram path is
the gather of finished tiles to the consumer rank, done as direct peer->root
sends (grouped ncclSend/ncclRecv through ``torch.distributed``; over xGMI every
peer has its own link to the root, a ring would be bound by one link).  The
Welch PSD needs one all-reduce of nfft floats.

One process per GPU; ``torch.distributed`` backend "nccl" is RCCL on ROCm,
"gloo" is used by the CPU tests (tests/test_dist.py).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_lines(total_lines: int, world: int, rank: int) -> Tuple[int, int]:
    """Half-open line range of ``rank``; ranges tile [0, total_lines) in order."""
    if not (0 <= rank < world):
        raise ValueError("rank %d of %d" % (rank, world))
    return rank * total_lines // world, (rank + 1) * total_lines // world


def shard_span(l0: int, l1: int, nfft: int, hop: int) -> Tuple[int, int]:
    """(first_sample, n_samples) a rank needs for lines [l0, l1), halo included."""
    if l1 <= l0:
        return l0 * hop, 0
    return l0 * hop, (l1 - l0 - 1) * hop + nfft


def total_lines(n_samples: int, nfft: int, hop: int) -> int:
    """Whole lines in a recording of n_samples (MainController.java:987 range test)."""
    return 0 if n_samples < nfft else (n_samples - nfft) // hop + 1


def gather_tiles(tile: torch.Tensor, total: int, nfft: int, dst: int = 0,
                 group: Optional[dist.ProcessGroup] = None) -> Optional[torch.Tensor]:
    """Gather the per-rank tiles ``[lines_r, nfft]`` into ``[total, nfft]`` on ``dst``.

    Tiles may have different heights (L not divisible by R): the root posts one
    receive per peer straight into the right rows of the result, every peer
    posts one send -- a grouped send/recv, no padding, no staging copy.
    """
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    l0, l1 = shard_lines(total, world, rank)
    if tile.shape != (l1 - l0, nfft):
        raise ValueError("rank %d tile is %s, expected (%d, %d)" % (rank, tuple(tile.shape), l1 - l0, nfft))
    tile = tile.contiguous()
    if rank == dst:
        out = torch.empty((total, nfft), dtype=tile.dtype, device=tile.device)
        out[l0:l1].copy_(tile)
        ops = []
        for r in range(world):
            if r == dst:
                continue
            a, b = shard_lines(total, world, r)
            if b > a:
                ops.append(dist.P2POp(dist.irecv, out[a:b], r, group))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return out
    if l1 > l0:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, tile, dst, group)]):
            w.wait()
    return None


def sharded_waterfall(compute_tile: Callable[[int, int], torch.Tensor], total: int, nfft: int,
                      gather_to: Optional[int] = 0, group: Optional[dist.ProcessGroup] = None):
    """Run ``compute_tile(l0, l1)`` for this rank's line range and gather.

    ``compute_tile`` returns the ``[l1 - l0, nfft]`` tile of lines l0..l1-1 on the
    rank's device (on a GPU: ``SpectralService.compute_waterfall`` over the
    rank's resident span).  With ``gather_to=None`` the tiles stay distributed
    (what a renderer that decimates first wants) and the local tile is returned.
    """
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    l0, l1 = shard_lines(total, world, rank)
    tile = compute_tile(l0, l1)
    if gather_to is None:
        return tile
    return gather_tiles(tile, total, nfft, gather_to, group)


def sharded_welch(partial_power: Callable[[int, int], torch.Tensor], n_seg: int, norm: float,
                  group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Welch PSD over segments sharded across ranks.

    ``partial_power(s0, s1)`` returns sum over segments s0..s1-1 of |FFT(w x_s)|^2
    (nfft values, this rank's device); one all-reduce(sum) of nfft floats
    combines the ranks, then every rank scales by ``norm / n_seg``.
    """
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    s0, s1 = shard_lines(n_seg, world, rank)
    acc = partial_power(s0, s1).to(torch.float64)
    dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return acc * (norm / n_seg)
