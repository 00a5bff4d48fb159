"""Multi-GPU sharding of a batch: one process per GPU, contiguous input-index ranges.

The walk has no data-path exchange -- inputs are independent and the DFA table
is replicated -- so the only collective is the gather of the per-shard accept
bitmaps (and, on request, end states) over RCCL (backend "nccl" on ROCm; "gloo"
in the CPU tests).  Shards are multiples of 64 inputs so bitmap words never
straddle two ranks.
"""
from __future__ import annotations


def shard_range(n_total: int, rank: int, world: int, align: int = 64):
    """Contiguous [first, first+count) of rank `rank`; every shard but the last is a
    multiple of `align` inputs, the sizes differ by less than 2*`align`."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    units = (n_total + align - 1) // align
    base, extra = divmod(units, world)
    first_u = rank * base + min(rank, extra)
    count_u = base + (1 if rank < extra else 0)
    first = min(first_u * align, n_total)
    last = min((first_u + count_u) * align, n_total)
    return first, last - first


def bitmap_words(count: int) -> int:
    return (count + 63) // 64


def gather_bitmap(local_words, world: int, group=None):
    """all_gather of equally sized per-rank bitmap tensors -> one tensor in rank order."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local_words
    out = torch.empty(local_words.numel() * world, dtype=local_words.dtype, device=local_words.device)
    dist.all_gather_into_tensor(out, local_words, group=group)
    return out


def gather_bitmap_ragged(local_words, counts, group=None):
    """Shards of different sizes (shard_range): pad to the largest, gather, trim."""
    import torch
    import torch.distributed as dist
    world = len(counts)
    if world == 1:
        return local_words
    words = [bitmap_words(c) for c in counts]
    mx = max(words)
    pad = torch.zeros(mx, dtype=local_words.dtype, device=local_words.device)
    pad[: local_words.numel()] = local_words
    out = torch.empty(mx * world, dtype=local_words.dtype, device=local_words.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * mx: r * mx + words[r]] for r in range(world)])


def assign_by_dfa(costs, world: int):
    """Many-DFA submissions shard BY DFA: job q goes to rank assign_by_dfa(costs, world)[q] -- largest cost
    first, each to the rank with the least work so far, ties to the lower rank.  The same rule as the C front's
    fsm_hip_multi_assign (tests/test_dist.py holds them equal), so every rank computes the same split without talking."""
    order = sorted(range(len(costs)), key=lambda q: -int(costs[q]))      # stable: equal costs keep their order
    load = [0] * world
    out = [0] * len(costs)
    for q in order:
        best = min(range(world), key=lambda g: (load[g], g))
        out[q] = best
        load[best] += int(costs[q]) or 1
    return out


def job_cost(n_lines: int, n_bytes: int) -> int:
    """what fsm_hip_node_exec_multi charges a job: its text bytes + 64 per line"""
    return int(n_bytes) + 64 * int(n_lines)
