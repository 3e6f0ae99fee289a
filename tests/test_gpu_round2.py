"""Round-2 GPU tests (run with -m gpu on an MI355X): the heavy-read pass with its dense candidate windows on a
repeat-rich genome, the limits the device-resident API enforces, the RCCL statistics reduce of the C ABI and
bench.py's own multi-rank launch path."""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest

import refio
from test_harness_cpu import assert_best_equal

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def wa():
    import walt_amd
    assert walt_amd.device_count() >= 1, "no HIP device: the walt_amd hot path has no CPU fallback"
    return walt_amd


def _mutate(rng, s, d):
    out = list(s)
    for i, c in enumerate(out):
        if rng.random() < d:
            out[i] = rng.choice([x for x in "ACGT" if x != c])
    return "".join(out)


def _revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


@pytest.fixture(scope="module")
def repeat_case(scratch):
    """Four sequences with the repeat classes of the hg19-like benchmark genome in miniature: a family of 260
    near-identical copies (regions of hundreds of candidates -> whole runs with dense records, both orientations),
    a family of 40 older copies (mid-size regions), 24 copies at 3 % (regions of 17-30: the own-lane batches and
    the cooperative path without records), a tandem array, and 6,000 exact copies of a 70-mer (regions beyond -b)."""
    rng = random.Random(77)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    young, old, mid, sat, tiny = rnd(220), rnd(260), rnd(180), rnd(171), rnd(70)
    seqs = []
    for c in range(4):
        parts = []
        for _ in range(65):
            parts.append(rnd(rng.randrange(150, 400)))
            u = _mutate(rng, young, 0.01)
            parts.append(u if rng.random() < 0.5 else _revcomp(u))
        for _ in range(10):
            parts.append(rnd(rng.randrange(100, 300)))
            parts.append(_mutate(rng, old, 0.08))
        for _ in range(6):
            parts.append(rnd(rng.randrange(100, 300)))
            parts.append(_mutate(rng, mid, 0.03))
        parts.append(rnd(200))
        parts.append("".join(_mutate(rng, sat, 0.02) for _ in range(60)))
        parts.append(rnd(200))
        parts.append("".join(tiny + rng.choice("ACGT") for _ in range(1500)))
        parts.append(rnd(300))
        seqs.append(("rep%d" % c, "".join(parts)))
    fa = os.path.join(scratch, "repeats.fa")
    with open(fa, "w") as f:
        for nm, s in seqs:
            f.write(">%s\n%s\n" % (nm, s))
    path = os.path.join(scratch, "repeats.dbindex")
    assert refio.harness().walt_makedb(fa.encode(), path.encode(), 4) == 0
    return seqs, refio.DbIndex(path)


def _reads(rng, seqs, n, conv, lens):
    a, b = ("C", "T") if conv == "CT" else ("G", "A")
    out = []
    while len(out) < n:
        _, g = seqs[rng.randrange(len(seqs))]
        L = rng.choice(lens)
        p = rng.randrange(0, len(g) - L)
        s = g[p:p + L]
        if rng.random() < 0.5:
            s = _revcomp(s)
        s = "".join(b if (c == a and rng.random() < 0.95) else c for c in s)
        s = _mutate(rng, s, 0.01)
        out.append(s)
    return out


@pytest.mark.parametrize("win", ["1", "0"])
def test_gpu_heavy_pass_and_dense_windows_vs_oracle(wa, repeat_case, win, monkeypatch):
    """Large regions through the heavy pass: with dense candidate windows (WALT_AMD_WIN default) and with the
    genome-gather fallback (WALT_AMD_WIN=0); 100-base reads (one record table), 150-base reads (both tables),
    mixed lengths incl. reads too long for the records; -b below and above the region sizes; both conversions."""
    seqs, db = repeat_case
    monkeypatch.setenv("WALT_AMD_WIN", win)
    idx = wa.Index.open(db.path, device=0, strands=wa.STRANDS_ALL, dir_bits=-1)
    if win == "1":
        assert idx.window_entries(0) > 1000 and idx.window_entries(2) > 1000
    else:
        assert idx.window_entries(0) == 0
    rng = random.Random(5)
    cases = [("CT", [100], 6, 5000), ("CT", [100], 6, 100), ("GA", [100], 4, 5000), ("CT", [150], 10, 5000),
             ("GA", [150], 10, 300), ("CT", [60, 100, 111, 112, 128, 150, 160, 170, 200], 6, 5000), ("CT", [100], 6, 20000)]
    for conv, lens, m, b in cases:
        reads = _reads(rng, seqs, 2500, conv, lens)
        want, work = refio.oracle_se(db, reads, ag=conv == "GA", max_mm=m, b=b)
        got, st = idx.map_se_batch(*wa.pack_reads(reads), ag_wildcard=conv == "GA", max_mismatches=m, b=b)
        assert_best_equal(got, want, "%s %s m=%d b=%d win=%s" % (conv, lens, m, b, win))
        assert int(st["big_regions"]) > 0 or b < 50
        # the kernels verify a superset of the reference's probes on the '-' strand (map_se.hip), never fewer
        assert int(st["candidates"]) >= int(work["cands"])
    # paired-end over the same repeats: the list kernels take the same cooperative path
    r1 = _reads(rng, seqs, 1200, "CT", [100])
    r2 = _reads(rng, seqs, 1200, "GA", [100])
    for k in (5, 50):
        want, _, _ = refio.oracle_pe(db, r1, r2, max_mm=6, b=5000, top_k=k, frag_range=1000)
        res, _ = idx.map_pe_batch(*wa.pack_reads(r1), *wa.pack_reads(r2), max_mismatches=6, top_k=k)
        for f in ("best_times", "frag_len", "best_i", "best_j", "pair_mm"):
            assert np.array_equal(res[f], want[f]), (k, f)
        assert_best_equal(res["m1"], want["m1"], "pair m1 k=%d" % k)
        assert_best_equal(res["m2"], want["m2"], "pair m2 k=%d" % k)
    idx.close()


@pytest.mark.parametrize("mode", ["mono_list", "small_state"])
def test_gpu_staged_paths_and_their_fallbacks_vs_oracle(wa, repeat_case, mode, monkeypatch):
    """The staged heavy pass (map_se.hip k_se_verify) and the staged paired-end path (map_pe.hip k_pe_stage /
    k_pe_verify / k_pe_push) are the default and run in every other test; here
    mono_list:   the one-kernel heavy pass and the list-kernel paired-end path they replaced (kept for comparison),
    small_state: the staged paths with tiny state -- the heavy list in eight chunks, the staged paired-end reads in
                 four rounds of 128 with the rest handed to the list kernel -- so that chunking, rounds, stage lists
                 and the fallback are all exercised on a batch of a few thousand reads.
    Same records as the oracle in every mode."""
    seqs, db = repeat_case
    idx = wa.Index.open(db.path, device=0, strands=wa.STRANDS_ALL, dir_bits=-1)
    if mode == "mono_list":
        idx.set_option("se_heavy_mono", 1)
        idx.set_option("pe_mode", 1)
    else:
        idx.set_option("se_heavy_chunk", 320)
        idx.set_option("pe_stage_cap", 128)
        idx.set_option("pe_rounds", 4)  # (a small index leaves the device roomy: one round would be the default)
    rng = random.Random(11)
    for conv, lens, m, b in (("CT", [100], 6, 5000), ("GA", [150], 10, 300), ("CT", [60, 100, 128, 150, 200], 6, 5000)):
        reads = _reads(rng, seqs, 2500, conv, lens)
        want, _ = refio.oracle_se(db, reads, ag=conv == "GA", max_mm=m, b=b)
        got, st = idx.map_se_batch(*wa.pack_reads(reads), ag_wildcard=conv == "GA", max_mismatches=m, b=b)
        assert_best_equal(got, want, "%s %s m=%d b=%d mode=%s" % (conv, lens, m, b, mode))
    r1 = _reads(rng, seqs, 1500, "CT", [100])
    r2 = _reads(rng, seqs, 1500, "GA", [100])
    for k in (5, 50, 300):
        want, _, _ = refio.oracle_pe(db, r1, r2, max_mm=6, b=5000, top_k=k, frag_range=1000)
        res, _ = idx.map_pe_batch(*wa.pack_reads(r1), *wa.pack_reads(r2), max_mismatches=6, top_k=k)
        for f in ("best_times", "frag_len", "best_i", "best_j", "pair_mm"):
            assert np.array_equal(res[f], want[f]), (mode, k, f)
        assert_best_equal(res["m1"], want["m1"], "pair m1 k=%d %s" % (k, mode))
        assert_best_equal(res["m2"], want["m2"], "pair m2 k=%d %s" % (k, mode))
    idx.close()


def test_gpu_profile_detail_accounts_for_the_mapping_time(wa, repeat_case):
    """walt_profile_detail (bench.py's roofline.by_kernel): the four kernel groups of a single-end call, timed by
    events between them, add up to what walt_profile_last reports for the mapping kernels."""
    seqs, db = repeat_case
    idx = wa.Index.open(db.path, device=0, strands=wa.STRANDS_CT, dir_bits=-1)
    idx.profile_enable(True)
    reads = _reads(random.Random(3), seqs, 4000, "CT", [100])
    idx.map_se_batch(*wa.pack_reads(reads), max_mismatches=6, b=5000)
    pack_ms, map_ms = idx.profile_last()
    detail = idx.profile_detail()
    assert len(detail) == 4 and all(x >= 0 for x in detail)
    assert detail[0] > 0 and detail[1] > 0 and detail[2] > 0  # pass 1, heavy stages, verifier all ran on this genome
    assert 0.5 * map_ms <= sum(detail) <= 1.05 * map_ms + 0.05, (detail, map_ms)
    idx.profile_enable(False)
    idx.close()


def test_gpu_device_api_refuses_reads_beyond_max_read_len(wa, g1_index_path, g1_db):
    """with max_read_len = 100 the 7-word kernels would take a 112-base read silently and the 2-bit
    conversion would run past the workspace.  Now such reads are refused in the kernels (record left as
    initialised), nothing is converted beyond n x max_read_len bytes, and walt_batch_check says WALT_EINVAL."""
    import torch
    idx = wa.Index.open(g1_index_path, device=0, strands=wa.STRANDS_CT)
    _, seqs, _ = next(refio.load_fastq_batches(os.path.join(refio.GOLDEN, "se_ct.fastq"), 10 ** 7))
    short = [s for s in seqs if len(s) == 100][:400]
    assert len(short) >= 100
    long_read = (short[0] + short[1])[:112]
    dev = torch.device("cuda:0")

    def run(reads, n_claim_len):
        bases, offsets = wa.pack_reads(reads)
        n = len(reads)
        d_bases = torch.from_numpy(bases).to(dev)
        d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
        d_out = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        d_stats = torch.zeros(4, dtype=torch.int64, device=dev)
        ws_bytes = wa.lib().walt_se_workspace_bytes(n, n_claim_len)
        guard = 1 << 16
        d_ws = torch.full((ws_bytes + guard,), 0xA5, dtype=torch.uint8, device=dev)  # guard bytes behind the workspace
        stream = torch.cuda.current_stream().cuda_stream
        idx.map_se_batch_device(d_bases.data_ptr(), d_off.data_ptr(), n, n_claim_len, d_out.data_ptr(), d_stats.data_ptr(),
                                d_ws.data_ptr(), ws_bytes, stream=stream)
        torch.cuda.synchronize()
        assert bool((d_ws[ws_bytes:] == 0xA5).all()), "the call wrote behind its workspace"
        return d_out.cpu().numpy().view(wa.best_match_dtype), d_ws, stream

    # one read longer than the caller said, in the middle of the batch
    reads = short[:50] + [long_read] + short[50:100]
    got, d_ws, stream = run(reads, 100)
    with pytest.raises(wa.WaltError) as ei:
        idx.check_batch(d_ws.data_ptr(), stream)
    assert ei.value.code == wa.WALT_EINVAL
    assert got[50]["times"] == 0 and got[50]["mismatch"] == 6  # refused: left as initialised
    # reads behind it lie (partly) beyond n x max_read_len bytes of the stream: refused too, never read out of bounds;
    # reads in front of it are unaffected
    want, _ = refio.oracle_se(g1_db, short[:50])
    assert_best_equal(got[:50], want, "reads in front of the over-long read")
    # every read 4 bases longer than max_read_len: the batch does not fit the workspace at all
    reads = [(s + "ACGT") for s in short[:200]]
    got, d_ws, stream = run(reads, 100)
    with pytest.raises(wa.WaltError) as ei:
        idx.check_batch(d_ws.data_ptr(), stream)
    assert ei.value.code == wa.WALT_EINVAL
    assert int(got["times"].sum()) == 0
    idx.close()


def test_gpu_c_abi_stats_allreduce_world_of_one(wa):
    """walt_comm_unique_id / walt_comm_init / walt_stats_allreduce over RCCL with one rank (a one-GPU box cannot
    hold two: RCCL refuses two ranks on one device); comm == NULL is the identity."""
    uid = wa.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    c = wa.Comm(0, 0, 1, uid)
    assert (c.rank, c.world) == (0, 1)
    v = np.array([50_000_000, 44_831_378, 2_805_922, 2_362_700, 0, 2 ** 40 + 5], dtype=np.uint64)
    out = c.stats_allreduce(v)
    assert out.tolist() == v.tolist()
    big = np.arange(1015, dtype=np.uint64) * 3  # a paired-end block: 4 + 2 x 5 + frag_range + 1 counters
    assert c.stats_allreduce(big).tolist() == big.tolist()
    c.close()
    w = v.copy()
    assert wa.lib().walt_stats_allreduce(None, w.ctypes.data, w.size) == 0 and w.tolist() == v.tolist()
    with pytest.raises(wa.WaltError):
        wa.Comm(0, 3, 2, uid)  # rank outside the world


def _bench(args, env_extra=None, timeout=500):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, env=env, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "bench.py must print exactly one JSON line: %r" % p.stdout[-500:]
    return json.loads(lines[0])


SMALL = ["--genome-mbp", "30", "--reads", "300000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra"]


def test_gpu_bench_two_ranks_on_one_gpu_equal_two_single_runs():
    """`python bench.py --gpus 2` starts its two ranks itself (child torch.distributed.run; here both on GPU 0 over
    gloo, WALT_AMD_BENCH_SHARE_GPU=1) and the all-reduced statistics equal the sum of the two ranks' own runs."""
    share = {"WALT_AMD_BENCH_SHARE_GPU": "1", "WALT_AMD_WIN_RESERVE_GB": "2"}
    two = _bench(["--gpus", "2"] + SMALL, share)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    r0 = _bench(["--gpus", "1", "--seed-offset", "0"] + SMALL, share)
    r1 = _bench(["--gpus", "1", "--seed-offset", "1"] + SMALL, share)
    assert r0["n_gpus"] == 1
    for k in ("total", "unique", "ambiguous", "unmapped", "too_short"):
        assert two["mapping"][k] == r0["mapping"][k] + r1["mapping"][k], k
    assert two["mapping"]["total"] == 600000


def test_gpu_bench_two_ranks_paired_end_full_stats_vector():
    share = {"WALT_AMD_BENCH_SHARE_GPU": "1", "WALT_AMD_WIN_RESERVE_GB": "2"}
    args = ["--mode", "pe", "--genome-mbp", "20", "--reads", "100000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
            "--no-extra"]
    two = _bench(["--gpus", "2"] + args, share)
    r0 = _bench(["--gpus", "1", "--seed-offset", "0"] + args, share)
    r1 = _bench(["--gpus", "1", "--seed-offset", "1"] + args, share)
    assert two["n_gpus"] == 2 and two["mapping"]["pairs"] == 200000
    for k in ("pairs", "unique_pairs", "ambiguous_pairs", "unpaired"):
        assert two["mapping"][k] == r0["mapping"][k] + r1["mapping"][k], k
    for mate in ("mate1", "mate2"):
        for k, v in two["mapping"][mate].items():
            assert v == r0["mapping"][mate][k] + r1["mapping"][mate][k], (mate, k)
    # the fragment-length histogram is part of the reduced vector: its mean is the pair-weighted mean of the ranks'
    w0, w1 = r0["mapping"]["unique_pairs"], r1["mapping"]["unique_pairs"]
    mean = (r0["mapping"]["frag_len_mean"] * w0 + r1["mapping"]["frag_len_mean"] * w1) / (w0 + w1)
    assert abs(two["mapping"]["frag_len_mean"] - mean) < 1e-6


def test_gpu_bench_refuses_a_world_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "--gpus 2" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_gpu_mapping_calls_ignore_the_former_environment_knobs(scratch, monkeypatch):
    """Until round 4 the mapping calls read a dozen environment variables -- one of them (WALT_AMD_ABLATE) made the product
    library return wrong mappings.  They are options of an index now (walt_index_set_option) and diagnostics live in
    libwalt_amd_diag.so only: with every former knob set in the environment the default library gives the oracle's records,
    single-end and paired-end."""
    import random

    import walt_amd as wa
    from test_harness_cpu import assert_best_equal, make_random_case, sample_reads
    seqs, db = make_random_case(41, 150, scratch)
    rng = random.Random(41)
    reads = sample_reads(rng, seqs, 2000, "CT")
    want, _ = refio.oracle_se(db, reads, ag=False, max_mm=6, b=5000)
    wantp, _, _ = refio.oracle_pe(db, reads[:800], reads[:800], max_mm=6, b=5000, top_k=50, frag_range=1000)
    for k, v in (("WALT_AMD_ABLATE", "7"), ("WALT_AMD_STAMPS", "1"), ("WALT_AMD_SYNC_DEBUG", "1"), ("WALT_AMD_HEAVY", "mono"),
                 ("WALT_AMD_SE_PIPE", "0"), ("WALT_AMD_PE", "list"), ("WALT_AMD_PE_SERIAL", "1"), ("WALT_AMD_SMALL_HEAPS", "1"),
                 ("WALT_AMD_GRID", "64"), ("WALT_AMD_HEAVY_CHUNK", "64"), ("WALT_AMD_PE_CHUNK", "128"),
                 ("WALT_AMD_PE_STAGE_CAP", "64"), ("WALT_AMD_DEFER_MIN", "0"), ("WALT_AMD_LIT_SIDE", "0")):
        monkeypatch.setenv(k, v)
    idx = wa.Index.open(db.path, device=0, strands=wa.STRANDS_ALL)
    got, _ = idx.map_se_batch(*wa.pack_reads(reads), ag_wildcard=False, max_mismatches=6, b=5000)
    assert_best_equal(got, want, "environment knobs set")
    res, _ = idx.map_pe_batch(*wa.pack_reads(reads[:800]), *wa.pack_reads(reads[:800]), max_mismatches=6, top_k=50)
    for f in ("best_times", "frag_len", "pair_mm"):
        assert np.array_equal(res[f], wantp[f]), f
    idx.close()
