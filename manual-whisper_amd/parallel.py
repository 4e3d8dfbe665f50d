"""Data-parallel transcription of one recording over the ranks of a node.

30 s chunks carry no state from one to the next, so rank r takes a contiguous block of chunks,
transcribes it on its own GPU with its own full copy of the weights, and the only exchange is ONE
all_gather of fixed-stride int32 rows ``[length, id_0 .. id_{max_new-1}]`` (RCCL over xGMI with the
``nccl`` backend; ``gloo`` on CPU for tests).  Every rank ends up with every chunk's ids in order.
No tensor parallelism, no all-reduce.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [a, b) of rank `rank`; blocks differ in size by at most one."""
    base, extra = divmod(n_items, world)
    a = rank * base + min(rank, extra)
    return a, a + base + (1 if rank < extra else 0)


def pack_rows(ids: Sequence[np.ndarray], max_new: int, rows: int) -> np.ndarray:
    """Variable-length id lists -> int32 [rows, 1 + max_new]; unused rows have length -1."""
    out = np.full((rows, 1 + max_new), -1, np.int32)
    for i, t in enumerate(ids):
        t = np.asarray(t, np.int32)[:max_new]
        out[i, 0] = len(t)
        out[i, 1:1 + len(t)] = t
    return out


def unpack_rows(rows: np.ndarray) -> List[np.ndarray]:
    return [r[1:1 + r[0]].copy() for r in rows if r[0] >= 0]


def broadcast_from_rank0(obj):
    """Rank 0's (small, picklable) object on every rank: the window list of a recording.  A world of one returns it."""
    try:
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
    except ImportError:
        on = False
    if not on or dist.get_world_size() == 1:
        return obj
    box = [obj if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def transcribe_sharded(transcribe_chunks: Callable[[np.ndarray], List[np.ndarray]], chunks, max_new: int,
                       device=None, n_items: int = None) -> List[np.ndarray]:
    """Run ``transcribe_chunks`` on this rank's share of the chunks and gather everyone's ids.

    ``chunks``: the array of all chunks, or (with ``n_items``) a callable ``(a, b) -> chunks[a:b]`` so that a rank
    only ever materialises its own share.  Works without torch.distributed initialised (world of one).
    ``device``: where the gathered tensor lives (a cuda device for nccl, None/cpu for gloo).
    """
    take = chunks if callable(chunks) else (lambda a, b: chunks[a:b])
    n = int(n_items) if n_items is not None else len(chunks)
    try:
        import torch
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
    except ImportError:
        on = False
    if not on:
        return [np.asarray(t, np.int32) for t in transcribe_chunks(take(0, n))] if n else []
    rank, world = dist.get_rank(), dist.get_world_size()
    a, b = shard_range(n, rank, world)
    mine = transcribe_chunks(take(a, b)) if b > a else []
    per = (n + world - 1) // world                       # fixed stride so one all_gather suffices
    rows = torch.from_numpy(pack_rows(mine, max_new, per))
    if device is not None:
        rows = rows.to(device)
    parts = [torch.empty_like(rows) for _ in range(world)]
    dist.all_gather(parts, rows)
    g = torch.stack(parts).cpu().numpy()
    out: List[np.ndarray] = []
    for r in range(world):                                # contiguous blocks: rank order is chunk order
        out.extend(unpack_rows(g[r]))
    assert len(out) == n, (len(out), n)
    return out


def align_sharded(align_segments: Callable[[list], dict], segments: Sequence[dict]) -> dict:
    """Forced alignment of a recording's segments over the ranks (reference transcribe.py:130-131 runs it on one
    device): segments are independent, so rank r aligns the contiguous block ``shard_range`` gives it with
    ``align_segments(block) -> {"segments": [...], "word_segments": [...]}`` and ONE gather of the (small, picklable)
    blocks puts the whole result together, in order, on every rank.  A world of one just calls ``align_segments``."""
    segments = list(segments)
    try:
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
    except ImportError:
        on = False
    if not on or dist.get_world_size() == 1:
        return align_segments(segments)
    rank, world = dist.get_rank(), dist.get_world_size()
    a, b = shard_range(len(segments), rank, world)
    mine = align_segments(segments[a:b]) if b > a else {"segments": [], "word_segments": []}
    blocks = [None] * world
    dist.all_gather_object(blocks, mine)
    out = {"segments": [], "word_segments": []}
    for blk in blocks:                                    # contiguous blocks: rank order is segment order
        out["segments"] += blk["segments"]
        out["word_segments"] += blk["word_segments"]
    return out
