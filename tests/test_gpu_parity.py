"""GPU parity tests proper: the HIP path, called through the C ABI
(include/walt_amd.h via the ctypes binding), against (a) the committed golden
files written by the real reference binary and (b) the oracle restatement on
seeded random inputs -- bit-exact on every record field."""
import os
import random

import numpy as np
import pytest

import refio
from test_harness_cpu import assert_best_equal, make_random_case, sample_reads
from test_oracle_golden import META, PE_CASES, SE_CASES, check_against_golden, run_pe_case, run_se_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wa():
    import walt_amd
    assert walt_amd.device_count() >= 1, "no HIP device: the walt_amd hot path has no CPU fallback"
    return walt_amd


@pytest.fixture(scope="module")
def g1_dev(wa, scratch):
    path = os.path.join(scratch, "g1_prod.dbindex")
    wa.makedb(os.path.join(refio.GOLDEN, "g1.fa"), path, threads=4)
    idx = {D: wa.Index.open(path, device=0, strands=wa.STRANDS_ALL, dir_bits=D) for D in (-1, 27)}
    yield idx
    for i in idx.values():
        i.close()


def dev_se_mapper(idx):
    def mapper(seqs, ag, m, b):
        import walt_amd
        bases, offsets = walt_amd.pack_reads(seqs)
        recs, stats = idx.map_se_batch(bases, offsets, ag_wildcard=ag, max_mismatches=m, b=b)
        return recs, int(stats["too_short"])
    return mapper


@pytest.mark.parametrize("D", [-1, 27])
@pytest.mark.parametrize("case", SE_CASES)
def test_gpu_se_reproduces_reference_files(g1_db, g1_dev, case, D):
    check_against_golden(case, run_se_case(g1_db, case, dev_se_mapper(g1_dev[D])))


@pytest.mark.parametrize("case", PE_CASES)
def test_gpu_pe_reproduces_reference_files(wa, g1_db, g1_dev, case):
    idx = g1_dev[27]

    def mapper(s1, s2, m, b, k, L):
        b1, o1 = wa.pack_reads(s1)
        b2, o2 = wa.pack_reads(s2)
        res, stats, ranked = idx.map_pe_batch(b1, o1, b2, o2, max_mismatches=m, b=b, top_k=k, frag_range=L,
                                              want_ranked=True)
        return res, ranked, (int(stats[0]["too_short"]), int(stats[1]["too_short"]))
    check_against_golden(case, run_pe_case(g1_db, case, mapper))


def test_gpu_se_records_equal_oracle_exact_times(wa, g1_db, g1_dev):
    """SAM only shows times as 0/1/>=2; compare the full BestMatch records."""
    for fq, ag in (("se_ct.fastq", False), ("se_ga.fastq", True)):
        _, seqs, _ = next(refio.load_fastq_batches(os.path.join(refio.GOLDEN, fq), 10 ** 7))
        for m, b in ((6, 5000), (10, 5000), (3, 7), (0, 5000)):
            want, work = refio.oracle_se(g1_db, seqs, ag=ag, max_mm=m, b=b)
            for D in (-1, 27):
                bases, offsets = wa.pack_reads(seqs)
                got, stats = g1_dev[D].map_se_batch(bases, offsets, ag_wildcard=ag, max_mismatches=m, b=b)
                assert_best_equal(got, want, "%s m=%d b=%d D=%d" % (fq, m, b, D))
                assert int(stats["too_short"]) == int(work["too_short"])
                assert int(stats["candidates"]) >= 0


@pytest.mark.parametrize("pe_chunk", [0, 333])
def test_gpu_pe_ranked_lists_and_pairs_equal_oracle(wa, g1_db, g1_dev, pe_chunk, index_options):
    """pe_chunk forces the batch through several workspace passes (two-stream fork/join per pass)."""
    index_options(g1_dev[-1], pe_chunk=pe_chunk)
    _, s1, _ = next(refio.load_fastq_batches(os.path.join(refio.GOLDEN, "pe_1.fastq"), 10 ** 7))
    _, s2, _ = next(refio.load_fastq_batches(os.path.join(refio.GOLDEN, "pe_2.fastq"), 10 ** 7))
    b1, o1 = wa.pack_reads(s1)
    b2, o2 = wa.pack_reads(s2)
    for k, m, L in ((2, 6, 1000), (5, 6, 1000), (50, 6, 1000), (300, 10, 400)):
        want, (r1, n1, r2, n2), _ = refio.oracle_pe(g1_db, s1, s2, max_mm=m, b=5000, top_k=k, frag_range=L)
        got, stats, (g1, gn1, g2, gn2) = g1_dev[-1].map_pe_batch(b1, o1, b2, o2, max_mismatches=m, b=5000, top_k=k,
                                                                 frag_range=L, want_ranked=True)
        assert np.array_equal(gn1, n1) and np.array_equal(gn2, n2)
        for j in range(len(s1)):
            for a, b_, c in ((g1, r1, n1), (g2, r2, n2)):
                for f in ("genome_pos", "strand", "mismatch"):
                    assert np.array_equal(a[j][:c[j]][f], b_[j][:c[j]][f]), (k, j, f)
        for f in ("best_times", "frag_len", "best_i", "best_j", "pair_mm"):
            assert np.array_equal(got[f], want[f]), f
        assert_best_equal(got["m1"], want["m1"], "m1")
        assert_best_equal(got["m2"], want["m2"], "m2")


@pytest.mark.parametrize("seed,n_chrom", [(11, 80), (12, 300), (13, 10), (14, 1500), (15, 2500), (16, 4500)])  # above 1,023 sequences the LDS table holds every 2nd, 4th, 8th start (map_common.h ChromTab)
def test_gpu_random_genomes_vs_oracle(wa, scratch, seed, n_chrom):
    seqs, db = make_random_case(seed, n_chrom, scratch)
    rng = random.Random(seed * 31)
    reads_ct = sample_reads(rng, seqs, 3000, "CT")
    reads_ga = sample_reads(rng, seqs, 1500, "GA")
    want_ct, wct = refio.oracle_se(db, reads_ct, ag=False, max_mm=6, b=5000)
    want_ga, _ = refio.oracle_se(db, reads_ga, ag=True, max_mm=4, b=50)
    bct, oct_ = wa.pack_reads(reads_ct)
    bga, oga = wa.pack_reads(reads_ga)
    any_bad = 0
    for D in (24, 26, 31):
        idx = wa.Index.open(db.path, device=0, strands=wa.STRANDS_ALL, dir_bits=D)
        any_bad += sum(idx.bad_buckets(s) + idx.outliers(s) for s in range(4))
        got, st = idx.map_se_batch(bct, oct_, ag_wildcard=False, max_mismatches=6, b=5000)
        assert_best_equal(got, want_ct, "CT D=%d" % D)
        assert int(st["too_short"]) == int(wct["too_short"])
        got, _ = idx.map_se_batch(bga, oga, ag_wildcard=True, max_mismatches=4, b=50)
        assert_best_equal(got, want_ga, "GA D=%d" % D)
        if D == 26:
            for k in (2, 50):
                # any reads can serve as "mate 2": it is mapped with G->A on the _GA1x strands
                res, _, (g1, gn1, g2, gn2) = idx.map_pe_batch(bct, oct_, bct, oct_, max_mismatches=6, top_k=k,
                                                              want_ranked=True)
                for ag, gr, gn in ((False, g1, gn1), (True, g2, gn2)):
                    ro, no, _ = refio.oracle_pe_topk(db, reads_ct, ag, 6, 5000, k)
                    assert np.array_equal(gn, no)
                    for j in range(len(reads_ct)):
                        assert np.array_equal(gr[j][:no[j]]["genome_pos"], ro[j][:no[j]]["genome_pos"]), (k, j)
                        assert np.array_equal(gr[j][:no[j]]["mismatch"], ro[j][:no[j]]["mismatch"]), (k, j)
                want, _, _ = refio.oracle_pe(db, reads_ct, reads_ct, max_mm=6, b=5000, top_k=k, frag_range=1000)
                for f in ("best_times", "frag_len", "best_i", "best_j", "pair_mm"):
                    assert np.array_equal(res[f], want[f]), (k, f)
                assert_best_equal(res["m1"], want["m1"], "pair m1 k=%d" % k)
                assert_best_equal(res["m2"], want["m2"], "pair m2 k=%d" % k)
        idx.close()
    assert any_bad > 0


@pytest.mark.parametrize("blocks", [1, 3])
def test_gpu_stage_hands_reads_out_over_several_windows(wa, scratch, blocks):
    """k_se_stage hands a wavefront's part of the heavy list out read by read, from list entries fetched a window of 64
    ahead (map_se.hip): with the default grid a test's few thousand reads never leave the first window, so the
    persistent kernels run on `blocks` blocks here (option grid) -- every wavefront walks through several windows."""
    seqs, db = make_random_case(12, 300, scratch)
    rng = random.Random(977)
    reads = sample_reads(rng, seqs, 6000, "CT")
    want, _ = refio.oracle_se(db, reads, ag=False, max_mm=6, b=5000)
    bct, oct_ = wa.pack_reads(reads)
    idx = wa.Index.open(db.path, device=0, strands=wa.STRANDS_ALL, dir_bits=24)
    idx.set_option("grid", blocks)
    for pipe in (1, 0):
        idx.set_option("se_pipe", pipe)
        got, st = idx.map_se_batch(bct, oct_, ag_wildcard=False, max_mismatches=6, b=5000)
        assert_best_equal(got, want, "stage blocks=%d pipe=%d" % (blocks, pipe))
    idx.close()


def test_gpu_long_reads_all_word_widths(wa, scratch):
    """Reads of 129..1000 bp take the 10-, 16-, 32- and 64-word kernel instances."""
    rng = random.Random(5)
    g = "".join(rng.choice("ACGT") for _ in range(20000))
    fa = os.path.join(scratch, "long.fa")
    with open(fa, "w") as f:
        f.write(">L\n%s\n" % g)
    path = os.path.join(scratch, "long.dbindex")
    wa.makedb(fa, path, threads=2)
    db = refio.DbIndex(path)
    idx = wa.Index.open(path, device=0, strands=wa.STRANDS_CT)
    for L in (129, 150, 160, 161, 256, 257, 300, 512, 600, 998):
        reads = []
        for _ in range(64):
            p = rng.randrange(0, len(g) - L)
            s = g[p:p + L]
            if rng.random() < 0.5:
                s = refio.revcomp(s)
            s = "".join("T" if c == "C" else c for c in s)
            s = "".join(rng.choice("ACGT") if rng.random() < 0.01 else c for c in s)
            reads.append(s)
        want, _ = refio.oracle_se(db, reads, max_mm=15)
        got, _ = idx.map_se_batch(*wa.pack_reads(reads), max_mismatches=15)
        assert_best_equal(got, want, "L=%d" % L)
    idx.close()


def test_gpu_errors_and_edges(wa, g1_dev):
    idx = g1_dev[-1]
    out, stats = idx.map_se_batch(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert out.size == 0
    with pytest.raises(wa.WaltError) as ei:
        idx.map_se_batch(*wa.pack_reads(["ACGTNACGT" * 12]))
    assert ei.value.code == -4  # WALT_EBASE: the reference exits in getBits (util.hpp:117-119)
    with pytest.raises(wa.WaltError):
        idx.map_pe_batch(*wa.pack_reads(["A" * 50]), *wa.pack_reads(["A" * 50]), top_k=1)  # walt.cpp:245-246
    with pytest.raises(wa.WaltError):
        wa.Index.open("/nonexistent/x.dbindex")
    # all reads too short
    out, stats = idx.map_se_batch(*wa.pack_reads(["ACGT" * 5, "A" * 37]))
    assert int(stats["too_short"]) == 4 and out["times"].tolist() == [0, 0]


def test_gpu_device_pointer_api_with_torch(wa, g1_db, g1_dev):
    import torch
    idx = g1_dev[-1]
    _, seqs, _ = next(refio.load_fastq_batches(os.path.join(refio.GOLDEN, "se_ct.fastq"), 10 ** 7))
    bases, offsets = wa.pack_reads(seqs)
    n = len(seqs)
    dev = torch.device("cuda:0")
    d_bases = torch.from_numpy(bases).to(dev)
    d_off = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    max_len = int((offsets[1:] - offsets[:-1]).max())
    d_out = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    d_stats = torch.zeros(4, dtype=torch.int64, device=dev)
    d_ws = torch.empty(wa.lib().walt_se_workspace_bytes(n, max_len), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    # a workspace smaller than the call needs is refused, not overrun
    with pytest.raises(wa.WaltError) as ei:
        idx.map_se_batch_device(d_bases.data_ptr(), d_off.data_ptr(), n, max_len, d_out.data_ptr(), d_stats.data_ptr(),
                                d_ws.data_ptr(), d_ws.numel() - 1, stream=stream)
    assert ei.value.code == wa.WALT_EINVAL
    idx.map_se_batch_device(d_bases.data_ptr(), d_off.data_ptr(), n, max_len, d_out.data_ptr(), d_stats.data_ptr(),
                            d_ws.data_ptr(), d_ws.numel(), stream=stream)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy().view(wa.best_match_dtype)
    want, work = refio.oracle_se(g1_db, seqs)
    assert_best_equal(got, want, "device api")
    assert int(d_stats[0]) == int(work["too_short"])
    idx.check_batch(d_ws.data_ptr(), stream)
    # an invalid base cannot come back as the status of the asynchronous call: walt_batch_check reports it
    bad = bases.copy()
    bad[offsets[5] + 3] = ord("N")
    d_bad = torch.from_numpy(bad).to(dev)
    idx.map_se_batch_device(d_bad.data_ptr(), d_off.data_ptr(), n, max_len, d_out.data_ptr(), d_stats.data_ptr(),
                            d_ws.data_ptr(), d_ws.numel(), stream=stream)
    with pytest.raises(wa.WaltError) as ei:
        idx.check_batch(d_ws.data_ptr(), stream)
    assert ei.value.code == wa.WALT_EBASE


def test_gpu_pe_device_api_pipelined_passes(wa, g1_db, g1_dev, index_options):
    """Device-resident paired-end call forced through many passes: they alternate between the two pipeline
    slots (own workspace and streams) and must give the oracle's pair records."""
    import torch
    idx = g1_dev[-1]
    index_options(idx, pe_chunk=257)
    _, s1, _ = next(refio.load_fastq_batches(os.path.join(refio.GOLDEN, "pe_1.fastq"), 10 ** 7))
    _, s2, _ = next(refio.load_fastq_batches(os.path.join(refio.GOLDEN, "pe_2.fastq"), 10 ** 7))
    b1, o1 = wa.pack_reads(s1)
    b2, o2 = wa.pack_reads(s2)
    n = len(s1)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    d1, d2 = t(b1), t(b2)
    do1, do2 = t(o1.astype(np.int64)), t(o2.astype(np.int64))
    max_len = int(max((o1[1:] - o1[:-1]).max(), (o2[1:] - o2[:-1]).max()))
    for k, m in ((50, 6), (3, 6), (300, 10)):
        want, _, _ = refio.oracle_pe(g1_db, s1, s2, max_mm=m, b=5000, top_k=k, frag_range=1000)
        d_out = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
        d_stats = torch.zeros(8, dtype=torch.int64, device=dev)
        d_ws = torch.empty(idx.pe_workspace_bytes(n, max_len, k), dtype=torch.uint8, device=dev)
        assert d_ws.numel() >= wa.lib().walt_pe_workspace_bytes(n, max_len, k) or idx.get_option("pe_chunk")
        stream = torch.cuda.current_stream().cuda_stream
        with pytest.raises(wa.WaltError) as ei:  # a workspace smaller than the call needs is refused, not overrun
            idx.map_pe_batch_device(d1.data_ptr(), do1.data_ptr(), d2.data_ptr(), do2.data_ptr(), n, max_len,
                                    d_out.data_ptr(), d_stats.data_ptr(), d_ws.data_ptr(), 4096, stream=stream,
                                    max_mismatches=m, top_k=k)
        assert ei.value.code == wa.WALT_EINVAL
        for _ in range(2):  # twice: slot state (events, streams) is reused across calls
            idx.map_pe_batch_device(d1.data_ptr(), do1.data_ptr(), d2.data_ptr(), do2.data_ptr(), n, max_len,
                                    d_out.data_ptr(), d_stats.data_ptr(), d_ws.data_ptr(), d_ws.numel(), stream=stream,
                                    max_mismatches=m, top_k=k)
            idx.check_batch(d_ws.data_ptr(), stream)
            got = d_out.cpu().numpy().view(wa.pair_result_dtype)
            for f in ("best_times", "frag_len", "best_i", "best_j", "pair_mm"):
                assert np.array_equal(got[f], want[f]), (k, f)
            assert_best_equal(got["m1"], want["m1"], "m1 k=%d" % k)
            assert_best_equal(got["m2"], want["m2"], "m2 k=%d" % k)


def test_gpu_crowded_chromosome_ends(wa, scratch):
    """Same stress as tests/test_harness_cpu.py::test_harness_crowded_chromosome_ends, on the device."""
    import random as _r
    rng = _r.Random(99)
    unit = "".join(rng.choice("ACGT") for _ in range(600))
    seqs = []
    for i in range(300):
        a = rng.randrange(0, 300)
        L = rng.choice([38, 60, 90, 131, 150, 200, 260])
        s = list(unit[a:a + L])
        for _ in range(rng.randrange(0, 3)):
            k = rng.randrange(len(s))
            s[k] = rng.choice("ACGT")
        seqs.append(("u%d" % i, "".join(s)))
    seqs.append(("long", unit * 3))
    fa = os.path.join(scratch, "crowded_gpu.fa")
    with open(fa, "w") as f:
        for nm, s in seqs:
            f.write(">%s\n%s\n" % (nm, s))
    idxp = os.path.join(scratch, "crowded_gpu.dbindex")
    wa.makedb(fa, idxp, threads=4)
    db = refio.DbIndex(idxp)
    reads = []
    for _ in range(6000):
        L = rng.choice([38, 50, 75, 100, 140, 150])
        a = rng.randrange(0, 600 * 3 - L)
        s = (unit * 3)[a:a + L]
        if rng.random() < 0.5:
            s = refio.revcomp(s)
        s = "".join("T" if (c == "C" and rng.random() < 0.9) else c for c in s)
        s = "".join(rng.choice("ACGT") if rng.random() < 0.01 else c for c in s)
        reads.append(s)
    want, _ = refio.oracle_se(db, reads, max_mm=6, b=5000)
    want_b, _ = refio.oracle_se(db, reads, max_mm=3, b=40)
    bases, offs = wa.pack_reads(reads)
    for B in (-1, 28):
        idx = wa.Index.open(idxp, device=0, strands=wa.STRANDS_ALL, dir_bits=B)
        assert sum(idx.outliers(s) for s in range(4)) > 100
        got, _ = idx.map_se_batch(bases, offs, max_mismatches=6, b=5000)
        assert_best_equal(got, want, "crowded B=%d" % B)
        got, _ = idx.map_se_batch(bases, offs, max_mismatches=3, b=40)
        assert_best_equal(got, want_b, "crowded b=40 B=%d" % B)
        for k in (2, 50):
            _, _, (g1, gn1, g2, gn2) = idx.map_pe_batch(bases, offs, bases, offs, top_k=k, want_ranked=True)
            ro, no, _ = refio.oracle_pe_topk(db, reads, False, 6, 5000, k)
            assert np.array_equal(gn1, no)
            for j in range(len(reads)):
                assert np.array_equal(g1[j][:no[j]]["genome_pos"], ro[j][:no[j]]["genome_pos"]), (k, j)
        idx.close()


def test_gpu_directory_of_2_pow_32_slots(wa, g1_db, scratch):
    """dir_bits = 32: slot numbers run to 2^32 and are handled modulo 2^32 (core.h dir_top); the all-zero code
    prefix (poly-A seeds on the G->A strands, where A is the one-bit letter) is the slot that wraps.  Same
    records as the oracle, single-end on both strand pairs and paired-end."""
    path = os.path.join(scratch, "g1_prod.dbindex")
    if not os.path.exists(path):
        wa.makedb(os.path.join(refio.GOLDEN, "g1.fa"), path, threads=4)
    idx = wa.Index.open(path, device=0, strands=wa.STRANDS_ALL, dir_bits=32)
    assert idx.dir_bits == 32
    extra = ["A" * 100, "A" * 60 + "ACGTTGCA" * 5, "T" * 100, "G" * 38, "C" * 64, "AG" * 50, "A" * 38]
    for fq, ag in (("se_ct.fastq", False), ("se_ga.fastq", True)):
        _, seqs, _ = next(refio.load_fastq_batches(os.path.join(refio.GOLDEN, fq), 10 ** 7))
        seqs = list(seqs) + extra
        want, work = refio.oracle_se(g1_db, seqs, ag=ag, max_mm=6, b=5000)
        got, stats = idx.map_se_batch(*wa.pack_reads(seqs), ag_wildcard=ag, max_mismatches=6, b=5000)
        assert_best_equal(got, want, "%s dir_bits=32" % fq)
    _, s1, _ = next(refio.load_fastq_batches(os.path.join(refio.GOLDEN, "pe_1.fastq"), 300))
    _, s2, _ = next(refio.load_fastq_batches(os.path.join(refio.GOLDEN, "pe_2.fastq"), 300))
    s1, s2 = list(s1) + extra, list(s2) + extra[::-1]
    res, _ = idx.map_pe_batch(*wa.pack_reads(s1), *wa.pack_reads(s2))
    wantp, _, _ = refio.oracle_pe(g1_db, s1, s2)
    for f in ("best_times", "frag_len", "pair_mm"):
        assert np.array_equal(res[f], wantp[f]), f
    idx.close()


def test_gpu_without_fence_keys(wa, scratch, monkeypatch):
    """WALT_AMD_FENCE=0 (read when an index is opened): no fence keys -- what an index opened on a device with little room
    to spare looks like -- so long slots and the literal search's group bounds go through the k-ary search over the
    entries (core.h slot_kary_bounds).  Same records as the oracle on a genome with long slots and many chromosome ends."""
    monkeypatch.setenv("WALT_AMD_FENCE", "0")
    seqs, db = make_random_case(31, 400, scratch)
    rng = random.Random(31 * 31)
    reads = sample_reads(rng, seqs, 3000, "CT")
    want, _ = refio.oracle_se(db, reads, ag=False, max_mm=6, b=5000)
    for D in (24, 28):
        idx = wa.Index.open(db.path, device=0, strands=wa.STRANDS_ALL, dir_bits=D)
        got, _ = idx.map_se_batch(*wa.pack_reads(reads), ag_wildcard=False, max_mismatches=6, b=5000)
        assert_best_equal(got, want, "no fence keys D=%d" % D)
        res, _ = idx.map_pe_batch(*wa.pack_reads(reads[:1000]), *wa.pack_reads(reads[:1000]), max_mismatches=6, top_k=50)
        wantp, _, _ = refio.oracle_pe(db, reads[:1000], reads[:1000], max_mm=6, b=5000, top_k=50, frag_range=1000)
        for f in ("best_times", "frag_len", "pair_mm"):
            assert np.array_equal(res[f], wantp[f]), (D, f)
        idx.close()


def test_gpu_slot_table(wa, g1_db, scratch, monkeypatch):
    """WALT_AMD_TABLE=1: the opt-in direct-mapped slot table (single-entry slots inline, core.h
    StrandView::tab) gives the same records as the oracle, single-end and paired-end, at two directory depths."""
    monkeypatch.setenv("WALT_AMD_TABLE", "1")
    path = os.path.join(scratch, "g1_prod.dbindex")
    if not os.path.exists(path):
        wa.makedb(os.path.join(refio.GOLDEN, "g1.fa"), path, threads=4)
    _, s1, _ = next(refio.load_fastq_batches(os.path.join(refio.GOLDEN, "pe_1.fastq"), 400))
    _, s2, _ = next(refio.load_fastq_batches(os.path.join(refio.GOLDEN, "pe_2.fastq"), 400))
    wantp, _, _ = refio.oracle_pe(g1_db, s1, s2)
    for D in (24, 28):
        idx = wa.Index.open(path, device=0, strands=wa.STRANDS_ALL, dir_bits=D)
        plain_bytes = None
        for fq, ag in (("se_ct.fastq", False), ("se_ga.fastq", True)):
            _, seqs, _ = next(refio.load_fastq_batches(os.path.join(refio.GOLDEN, fq), 10 ** 7))
            want, _ = refio.oracle_se(g1_db, seqs, ag=ag, max_mm=6, b=5000)
            got, _ = idx.map_se_batch(*wa.pack_reads(seqs), ag_wildcard=ag, max_mismatches=6, b=5000)
            assert_best_equal(got, want, "%s slot table D=%d" % (fq, D))
        res, _ = idx.map_pe_batch(*wa.pack_reads(s1), *wa.pack_reads(s2))
        for f in ("best_times", "frag_len", "pair_mm"):
            assert np.array_equal(res[f], wantp[f]), f
        # the table is really there: 12 bytes per slot and strand on top of the plain index
        monkeypatch.setenv("WALT_AMD_TABLE", "0")
        plain = wa.Index.open(path, device=0, strands=wa.STRANDS_ALL, dir_bits=D)
        plain_bytes = plain.device_bytes
        plain.close()
        monkeypatch.setenv("WALT_AMD_TABLE", "1")
        assert idx.device_bytes - plain_bytes >= 4 * 12 * (1 << D)
        idx.close()


def test_gpu_filter_covers_every_dangerous_probe(wa_diag, scratch, monkeypatch):
    """In-kernel self-check of the DIAGNOSTIC build (libwalt_amd_diag.so: WALT_AMD_STAMPS=1 WALT_AMD_ABLATE=8 -- the
    product library reads neither): for every probe pass 1 issues, the exact test
    (core.h probe_is_dangerous) is evaluated beside the prefilter/Bloom decision; a probe that is dangerous but
    not flagged would silently take the key search.  Genomes with hundreds of chromosome ends; the count of
    such probes must be zero while thousands of probes are checked."""
    import ctypes
    wa = wa_diag
    monkeypatch.setenv("WALT_AMD_STAMPS", "1")
    monkeypatch.setenv("WALT_AMD_ABLATE", "8")
    checked = 0
    for seed, n_chrom in ((21, 300), (22, 1100)):
        seqs, db = make_random_case(seed, n_chrom, scratch)
        rng = random.Random(seed)
        reads = [r for r in sample_reads(rng, seqs, 4000, "CT") if len(r) <= 112]  # the 7-word pass-1 instance
        idx = wa.Index.from_host(db.lengths, db.genome, db.counter, db.index, chrom_names=db.names, device=0, diag=True)
        want, _ = refio.oracle_se(db, reads)
        got, _ = idx.map_se_batch(*wa.pack_reads(reads))
        assert_best_equal(got, want, "self-check run")  # results stay valid in this mode
        buf = (ctypes.c_ulonglong * 16)()
        assert wa.diag_lib().walt_profile_stamps(buf) == 0
        assert buf[15] == 0, "%d dangerous probes were not flagged by the filter" % buf[15]
        checked += buf[14]
        idx.close()
    assert checked > 5000


def test_gpu_pe_small_heaps_with_overflow_list(wa, scratch, monkeypatch):
    """The literal list kernel of the paired-end path switches to 8-slot heaps (64 reads per wavefront) when its
    list is long, and hands reads that collect more candidates to an overflow list mapped with full heaps.
    WALT_AMD_SMALL_HEAPS forces that mode.  Genome: 300 short chromosomes cut from one repeated sequence, so
    that most reads are deferred to the literal list (chromosome ends everywhere) AND collect dozens of
    candidates (every read occurs in many chromosomes): the overflow path carries most of the load.  Pair
    records and ranked lists must equal the oracle's."""
    rng = random.Random(4242)
    unit = "".join(rng.choice("ACGT") for _ in range(600))
    seqs = []
    for i in range(300):
        a = rng.randrange(0, 300)
        L = rng.choice([60, 90, 131, 150, 200, 260])
        s_ = list(unit[a:a + L])
        for _ in range(rng.randrange(0, 3)):
            s_[rng.randrange(len(s_))] = rng.choice("ACGT")
        seqs.append(("u%d" % i, "".join(s_)))
    seqs.append(("long", unit * 3))
    fa = os.path.join(scratch, "crowded_pe.fa")
    with open(fa, "w") as f:
        for nm, s_ in seqs:
            f.write(">%s\n%s\n" % (nm, s_))
    path = os.path.join(scratch, "crowded_pe.dbindex")
    wa.makedb(fa, path, threads=4)
    db = refio.DbIndex(path)
    s1, s2 = [], []
    big = unit * 3
    for _ in range(1500):
        flen = rng.randrange(120, 400)
        a = rng.randrange(0, len(big) - flen)
        frag = big[a:a + flen]
        if rng.random() < 0.5:
            frag = refio.revcomp(frag)
        frag = "".join("T" if (c == "C" and rng.random() < 0.9) else c for c in frag)
        L = rng.choice([50, 75, 100])
        s1.append(frag[:L])
        s2.append(refio.revcomp(frag)[:L])
    idx = wa.Index.open(path, device=0)
    idx.set_option("pe_small_heaps", 1)
    idx.set_option("pe_mode", 1)  # the list kernels (the staged path maps long literal lists itself)
    deferred_any = False
    for k in (50, 5, 300):
        res, stats, ranked = idx.map_pe_batch(*wa.pack_reads(s1), *wa.pack_reads(s2), top_k=k, want_ranked=True)
        wantp, wranked, _ = refio.oracle_pe(db, s1, s2, top_k=k)
        for f in ("best_times", "frag_len", "pair_mm", "best_i", "best_j"):
            assert np.array_equal(res[f], wantp[f]), (k, f)
        assert np.array_equal(ranked[1], wranked[1]) and np.array_equal(ranked[3], wranked[3])
        deferred_any = deferred_any or int(np.max(ranked[1])) > 8
    assert deferred_any, "the read set should produce lists longer than the small heaps"
    idx.close()
