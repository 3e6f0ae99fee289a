"""world_size-2 gloo test of the N>1 path's only collective: per-rank shards of a
read set are mapped independently (here by the oracle, on CPU) and the summed
statistics vector equals the single-process one; shard_range keeps output order."""
import os
import socket

import numpy as np
import pytest

import refio


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, index_path, fq, out_dir):
    import torch
    import torch.distributed as dist
    from walt_amd import dist as wd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    db = refio.DbIndex(index_path, strands=(0, 1))
    _, seqs, _ = next(refio.load_fastq_batches(fq, 10 ** 7))
    lo, hi = wd.shard_range(len(seqs), rank, world)
    recs, work = refio.oracle_se(db, seqs[lo:hi], threads=2)
    vec = wd.se_stats_vector(torch.from_numpy(recs["times"].astype(np.int64)), int(work["too_short"]))
    wd.allreduce_stats(vec)
    t = wd.allreduce_max(1.0 + rank)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), recs)
    if rank == 0:
        np.save(os.path.join(out_dir, "stats.npy"), vec.numpy())
        np.save(os.path.join(out_dir, "tmax.npy"), np.array([t]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_stats_allreduce_gloo(g1_index_path, g1_db, scratch):
    import torch.multiprocessing as mp
    from walt_amd import dist as wd
    fq = os.path.join(refio.GOLDEN, "se_ct.fastq")
    out_dir = os.path.join(scratch, "dist2")
    os.makedirs(out_dir, exist_ok=True)
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), g1_index_path, fq, out_dir), nprocs=world, join=True)
    _, seqs, _ = next(refio.load_fastq_batches(fq, 10 ** 7))
    want, work = refio.oracle_se(g1_db, seqs)
    t = want["times"]
    expect = [len(seqs), int((t == 1).sum()), int((t >= 2).sum()), int((t == 0).sum()), int(work["too_short"])]
    assert np.load(os.path.join(out_dir, "stats.npy")).tolist() == expect
    assert np.load(os.path.join(out_dir, "tmax.npy"))[0] == 2.0
    cat = np.concatenate([np.load(os.path.join(out_dir, "rank%d.npy" % r)) for r in range(world)])
    assert cat.tobytes() == want.tobytes()  # rank-order concatenation == single-process order


def test_shard_range_partitions():
    from walt_amd import dist as wd
    for n in (0, 1, 7, 8, 50_000_001):
        for w in (1, 2, 3, 8):
            spans = [wd.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
