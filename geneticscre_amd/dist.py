"""Multi-GPU exchange step of the path-join scorer: one process per GPU, torch.distributed (RCCL on ROCm).

Joined paths of one level are independent (reference src/join_base.cpp:230-250: workers pull uids from an atomic
counter), so a level shards into contiguous slices of joined-path ordinals with no data-path collective.  What
the reference merges under its mutex (merge_scores, src/methods.h:25-39) becomes, once per level:

* element-wise MAX all-reduce of the K f32 null maxima -- exact for any sharding because f32 max is associative
  and every rank already holds f32-rounded values;
* an all-gather of every rank's top-k table, merged identically on every rank.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from .uids import UidRelSet

SENTINEL = (-np.inf, -1, -1, 0, 0)   # Score() default, src/gcre_types.h:34-38


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    return (total * rank) // world, (total * (rank + 1)) // world


def shard_uids(uids: UidRelSet, begin: int, end: int) -> UidRelSet:
    """The join index restricted to joined-path ordinals [begin, end): same rows, trimmed counts/locations, so
    the (idx, loc) ids of every surviving path are unchanged."""
    first = uids.path_idx[:-1]
    last = uids.path_idx[1:]
    lo = np.clip(begin, first, last)
    hi = np.clip(end, first, last)
    count = (hi - lo).astype(np.int32)
    location = np.where(count > 0, uids.location + (lo - first), uids.location).astype(np.int64)
    return UidRelSet(uids.path_length, uids.src, uids.trg, count, location, uids.signs)


def merge_topk(rows: np.ndarray, top_k: int) -> np.ndarray:
    """rows [m, 5] = (score, src, trg, cases, ctrls) from all ranks -> the reference's result table
    (format_result, src/join_base.cpp:138-154): best top_k of {sentinel} U rows, ascending; equal scores are
    ordered by (src, trg), i.e. by joined-path ordinal (DESIGN.md "Ties")."""
    rows = rows[rows[:, 1] >= 0]                                   # every rank's sentinel / padding
    order = np.lexsort((rows[:, 2], rows[:, 1], -rows[:, 0]))      # score desc, src asc, trg asc
    best = rows[order[:top_k]]
    if len(best) < top_k:
        best = np.vstack([best, np.array([SENTINEL])])
    return best[::-1].copy()


def exchange_level(scores, src, trg, cases, ctrls, null_tensor, top_k: int, world: int, device=None):
    """All-reduce(MAX) ``null_tensor`` in place and all-gather + merge the top-k table.  Returns the merged
    [n, 5] table.  Works on any torch.distributed backend (nccl == RCCL on the GPU box, gloo in the CPU tests)."""
    import torch
    import torch.distributed as dist

    if world > 1 and null_tensor.numel() > 0:
        dist.all_reduce(null_tensor, op=dist.ReduceOp.MAX)
    mine = torch.full((top_k + 1, 5), -1.0, dtype=torch.float64)
    mine[:, 0] = float("-inf")
    rows = np.stack([np.asarray(scores, dtype=np.float64), src, trg, cases, ctrls], axis=1).astype(np.float64)
    mine[: len(rows)] = torch.from_numpy(rows)
    if device is not None:
        mine = mine.to(device)
    if world > 1:
        allrows = torch.empty((world * (top_k + 1), 5), dtype=torch.float64, device=mine.device)
        dist.all_gather_into_tensor(allrows, mine)
    else:
        allrows = mine
    return merge_topk(allrows.cpu().numpy(), top_k)


def agree_window(local_window: int, world: int, device=None) -> int:
    """One permutation window for every rank: the smallest any rank planned (MIN all-reduce).  Each rank sizes its
    window from ITS free device memory (gcre_plan_perm_window), and every (level, window) pair is one MAX all-reduce plus
    one all-gather: ranks that walked different window lists would issue collectives of different sizes and counts."""
    if world <= 1:
        return int(local_window)
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(local_window)], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())
