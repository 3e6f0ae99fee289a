"""world_size-2 gloo test of the N>1 path's only collective: per-rank shards of a
read set are mapped independently (here by the oracle, on CPU) and the summed
statistics vector equals the single-process one; shard_range keeps output order.
Also: the paired-end statistics block (walt_amd.dist.pe_stats_vector) against a plain restatement of how
ProcessPairedEndReads accumulates it, and bench.py's own launch path (`--gpus 2` starts two ranks) in its
CPU dry-run form."""
import json
import os
import socket
import subprocess
import sys

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


def _reference_pe_stats(recs, frag_range):
    """StatPairedReads as paired.cpp:515-547 fills it, one pair at a time."""
    pairs = [0, 0, 0, 0]
    mates = [[0, 0, 0, 0, 0], [0, 0, 0, 0, 0]]
    hist = [0] * (frag_range + 1)
    for r in recs:
        pairs[0] += 1
        if r["best_times"] == 1:
            pairs[1] += 1
            hist[int(r["frag_len"])] += 1          # paired.cpp:526
            continue
        pairs[2 if r["best_times"] >= 2 else 3] += 1
        for k, m in enumerate(("m1", "m2")):       # paired.cpp:546-547 -> StatInfoUpdate, mapping.cpp:318-327
            t = int(r[m]["times"])
            mates[k][0] += 1
            mates[k][1 if t == 1 else (2 if t >= 2 else 3)] += 1
    return pairs + mates[0] + mates[1] + hist


def _random_pair_records(rng, n, frag_range):
    import walt_amd
    recs = np.zeros(n, dtype=walt_amd.pair_result_dtype)
    recs["best_times"] = rng.choice([0, 1, 1, 1, 2, 5], size=n)
    recs["frag_len"] = np.where(recs["best_times"] == 1, rng.integers(1, frag_range + 1, size=n), 0)
    for m in ("m1", "m2"):
        recs[m]["times"] = np.where(recs["best_times"] == 1, 1, rng.choice([0, 1, 2, 3], size=n))
    return recs


def test_pe_stats_vector_is_the_reference_accumulation():
    from walt_amd import dist as wd
    rng = np.random.default_rng(3)
    for frag_range in (1000, 37):
        recs = _random_pair_records(rng, 5000, frag_range)
        words = recs.view(np.int32).reshape(-1, 16)
        got = wd.pe_stats_vector(words, frag_range, too_short1=6, too_short2=2).tolist()
        want = _reference_pe_stats(recs, frag_range)
        want[8], want[13] = 6, 2
        assert len(got) == wd.pe_stats_len(frag_range) == 4 + 10 + frag_range + 1
        assert got == want


def _pe_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from walt_amd import dist as wd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    recs = _random_pair_records(np.random.default_rng(11), 4001, 1000)
    lo, hi = wd.shard_range(len(recs), rank, world)
    words = torch.from_numpy(recs[lo:hi].view(np.int32).reshape(-1, 16).copy())
    vec = wd.allreduce_stats(wd.pe_stats_vector(words, 1000, too_short1=rank, too_short2=2 * rank))
    if rank == 0:
        np.save(os.path.join(out_dir, "pe_stats.npy"), vec.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_pe_stats_allreduce_gloo(scratch):
    import torch.multiprocessing as mp
    out_dir = os.path.join(scratch, "dist_pe")
    os.makedirs(out_dir, exist_ok=True)
    mp.spawn(_pe_worker, args=(2, _free_port(), out_dir), nprocs=2, join=True)
    want = _reference_pe_stats(_random_pair_records(np.random.default_rng(11), 4001, 1000), 1000)
    want[8], want[13] = 0 + 1, 0 + 2  # too_short of the two ranks
    assert np.load(os.path.join(out_dir, "pe_stats.npy")).tolist() == want


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` without a launcher: the parent starts torch.distributed.run as a child before
    anything touches a GPU; WALT_AMD_BENCH_DRYRUN=1 makes the ranks stop after the rendezvous, the world-size
    check and one all-reduce (gloo), so the launch path is covered on the CPU box."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WALT_AMD_BENCH_DRYRUN="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-1500:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["stats"] == [2, 1, 30]
    # an external launcher that started the wrong number of ranks is refused
    env.update(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=env, timeout=120)
    assert p.returncode != 0 and "--gpus 2" in p.stderr


def _cross_check_worker(rank, world, port, out_dir, break_rank):
    import torch.distributed as dist
    from walt_amd import dist as wd
    import walt_amd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == break_rank:  # this rank "cannot load librccl": the others must not be left waiting for it
        walt_amd.comm_available = lambda: False
    res = wd.c_abi_cross_check(0, np.array([1, 2, 3], dtype=np.uint64), [2, 4, 6])
    open(os.path.join(out_dir, "xcheck%d.txt" % rank), "w").write(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("break_rank", [0, 1])
def test_c_abi_cross_check_fails_collectively_not_by_hanging(scratch, break_rank):
    """walt_amd.dist.c_abi_cross_check (bench.py's walt_stats_allreduce check at N > 1): when ONE rank cannot use
    librccl -- or, on this GPU-less box, when rank 0 cannot make the id -- every rank gets "failed: ..." and goes on;
    nobody enters ncclCommInitRank or a broadcast alone (the decision must be collective)."""
    import torch.multiprocessing as mp
    out_dir = os.path.join(scratch, "xcheck%d" % break_rank)
    os.makedirs(out_dir, exist_ok=True)
    world = 2
    mp.spawn(_cross_check_worker, args=(world, _free_port(), out_dir, break_rank), nprocs=world, join=True)
    texts = [open(os.path.join(out_dir, "xcheck%d.txt" % r)).read() for r in range(world)]
    assert all(t.startswith("failed:") for t in texts), texts
