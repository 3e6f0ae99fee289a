"""Multi-GPU plumbing of the hot path: reads shard by contiguous blocks, every
rank holds a full index replica, and the ONLY collective is the final sum of the
mapping statistics (StatSingleReads, mapping.hpp:94-100; StatPairedReads incl. the
fragment-length histogram, paired.hpp:96-105) -- nccl (= RCCL over xGMI) on GPUs,
gloo in the CPU tests.  torch is imported lazily: the single-GPU binding does
not need it."""


def shard_range(n_total, rank, world_size):
    """Contiguous, order-preserving block of [0, n_total) for `rank` (concatenating
    the ranks' outputs in rank order reproduces the single-process output order)."""
    base, rem = divmod(int(n_total), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


SE_FIELDS = ("total_reads", "unique_mapped_reads", "ambiguous_mapped_reads", "unmapped_reads", "num_of_short_reads")
PE_FIELDS = ("total_read_pairs", "unique_mapped_pairs", "ambiguous_mapped_pairs", "unmapped_pairs")


def se_stats_vector(times, too_short):
    """times: per-read BestMatch.times (numpy or torch int tensor) -> the five
    StatSingleReads counters as StatInfoUpdate accumulates them (mapping.cpp:318-327)."""
    import torch
    t = torch.as_tensor(times)
    return torch.stack([torch.tensor(t.numel(), device=t.device), (t == 1).sum(), (t >= 2).sum(), (t == 0).sum(),
                        torch.tensor(int(too_short), device=t.device)]).to(torch.int64)


def allreduce_stats(vec, group=None):
    """Sum an int64 statistics vector over all ranks (no-op without a process group)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    return vec


def allreduce_max(value, device="cpu", group=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
