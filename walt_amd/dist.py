"""Multi-GPU plumbing of the hot path: reads shard by contiguous blocks, every
rank holds a full index replica, and the ONLY collective is the final sum of the
mapping statistics (StatSingleReads, mapping.hpp:94-100; StatPairedReads incl. the
fragment-length histogram, paired.hpp:96-105) -- nccl (= RCCL over xGMI) on GPUs,
gloo in the CPU tests; the same sum is available without torch through the C ABI
(walt_stats_allreduce, walt_amd.Comm).  torch is imported lazily: the single-GPU
binding does not need it."""


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


def pe_stats_len(frag_range):
    return 4 + 2 * 5 + int(frag_range) + 1


def pe_stats_vector(pair_words, frag_range, too_short1=0, too_short2=0):
    """StatPairedReads as ProcessPairedEndReads accumulates it (paired.hpp:96-105):
         [0:4]    total_read_pairs, unique, ambiguous, unmapped pairs          (paired.cpp:519-539)
         [4:9]    mate 1: total, unique, ambiguous, unmapped, too_short        (updated only for pairs
         [9:14]   mate 2                                                        without a unique best pair, 546-547)
         [14:]    fragment_len_count[0 .. frag_range]                          (paired.cpp:526)
    pair_words: the walt_pair_result records of a batch viewed as an [n, 16] int32 tensor (numpy or torch):
    m1 = words 0..3, m2 = 4..7 (times at +1), best_times = word 8, frag_len = word 9."""
    import torch
    w = torch.as_tensor(pair_words)
    dev = w.device
    bt = w[:, 8]
    uniq = bt == 1
    out = torch.zeros(pe_stats_len(frag_range), dtype=torch.int64, device=dev)
    out[0] = w.shape[0]
    out[1] = uniq.sum()
    out[2] = (bt >= 2).sum()
    out[3] = (bt == 0).sum()
    rest = ~uniq
    for k, (col, short) in enumerate(((1, too_short1), (5, too_short2))):
        t = w[:, col][rest]
        base = 4 + 5 * k
        out[base] = t.numel()
        out[base + 1] = (t == 1).sum()
        out[base + 2] = (t >= 2).sum()
        out[base + 3] = (t == 0).sum()
        out[base + 4] = int(short)
    fl = w[:, 9][uniq].to(torch.int64)
    if fl.numel():
        out[14:] = torch.bincount(fl.clamp(0, int(frag_range)), minlength=int(frag_range) + 1)
    return out


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


def c_abi_comm(device, group=None):
    """walt_amd.Comm (RCCL through the C ABI) for the ranks of the initialised torch.distributed group:
    rank 0 makes the unique id and torch.distributed carries it to the others (the out-of-band step
    walt_comm_unique_id asks for)."""
    import torch
    import torch.distributed as dist
    import walt_amd
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [walt_amd.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return walt_amd.Comm(device, rank, world, box[0])


def _all_min(flag, group=None):
    """MIN of a 0/1 flag over the ranks, carried by the job's own process group (gloo or nccl)."""
    import torch
    import torch.distributed as dist
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item())


def c_abi_cross_check(device, vec_local, expect, group=None):
    """The sum of `vec_local` over the ranks through the C ABI's own communicator (walt_stats_allreduce over RCCL),
    compared with `expect` (the same sum made by torch.distributed).  COLLECTIVE, and written so that a failure on
    some ranks cannot leave the others blocked: every step that can fail locally is followed by a MIN over the ranks
    of an ok flag on the job's process group, and only a unanimous yes enters the next blocking call; the id is
    broadcast unconditionally (None when rank 0 could not make it).  Returns "equal", "DIFFERENT: ..." or
    "failed: ...", the same string on every rank up to the local error text."""
    import torch.distributed as dist
    import walt_amd
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    err = ""
    uid = None
    ok = walt_amd.comm_available()
    if not ok:
        err = "librccl did not load on rank %d" % rank
    elif rank == 0:
        try:
            uid = walt_amd.comm_unique_id()
        except Exception as e:  # WaltError
            ok, err = False, "walt_comm_unique_id: %s" % e
    box = [uid]
    dist.broadcast_object_list(box, src=0, group=group)
    if not _all_min(ok and box[0] is not None, group):
        return "failed: %s" % (err or "another rank could not load librccl / make the id")
    comm = None
    try:
        comm = walt_amd.Comm(device, rank, world, box[0])
    except Exception as e:
        err = "walt_comm_init: %s" % e
    if not _all_min(comm is not None, group):
        if comm is not None:
            comm.close()
        return "failed: %s" % (err or "walt_comm_init failed on another rank")
    res = None
    try:
        res = [int(x) for x in comm.stats_allreduce(vec_local).tolist()]
    except Exception as e:
        err = "walt_stats_allreduce: %s" % e
    comm.close()
    if not _all_min(res is not None, group):
        return "failed: %s" % (err or "walt_stats_allreduce failed on another rank")
    return "equal" if res == [int(x) for x in expect] else "DIFFERENT: %s" % res
