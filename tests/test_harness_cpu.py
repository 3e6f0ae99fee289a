"""CPU checks of the per-lane kernel logic (core.h / index_core.h compiled with
g++ by tests/host_harness.cpp) against the oracle: derived index structures,
directory + key search vs the literal LowerBound/UpperBound narrowing, packed
reads, masked mismatch counting, BestMatch fold, libstdc++ heap emulation and
the pair merge.  The HIP kernels use exactly these functions per lane."""
import os
import random

import numpy as np
import pytest

import refio
from test_oracle_golden import META, check_against_golden, run_pe_case, run_se_case


def assert_best_equal(got, want, what=""):
    for f in ("genome_pos", "times", "strand", "mismatch"):
        bad = np.nonzero(got[f] != want[f])[0]
        assert bad.size == 0, "%s field %s differs at %s: got %s want %s" % (
            what, f, bad[:5], got[f][bad[:5]], want[f][bad[:5]])


@pytest.fixture(scope="module")
def g1_harness(g1_db):
    hs = {D: refio.HarnessIndex(g1_db, D) for D in (24, 26)}
    yield hs
    for h in hs.values():
        h.close()


@pytest.mark.parametrize("D", [24, 26])
@pytest.mark.parametrize("case", ["se_sam_au", "se_mr_au", "se_ag_sam_au", "se_sam_au_b2", "se_sam_au_N100",
                                  "se_sam_au_m10"])
def test_harness_se_reproduces_reference_files(g1_db, g1_harness, case, D):
    h = g1_harness[D]
    check_against_golden(case, run_se_case(g1_db, case, lambda s, ag, m, b: h.map_se(s, ag, m, b)))


@pytest.mark.parametrize("case", ["pe_sam_au", "pe_mr_au", "pe_sam_au_k3", "pe_sam_au_k300", "pe_sam_au_m10_b20",
                                  "pe_sam_au_L200"])
def test_harness_pe_reproduces_reference_files(g1_db, g1_harness, case):
    h = g1_harness[26]

    def mapper(s1, s2, m, b, k, L):
        r1, n1, t1 = h.pe_topk(s1, False, m, b, k)
        r2, n2, t2 = h.pe_topk(s2, True, m, b, k)
        res = h.pe_merge(r1, n1, r2, n2, s1, s2, k, L, m)
        return res, (r1, n1, r2, n2), (t1, t2)
    check_against_golden(case, run_pe_case(g1_db, case, mapper))


def test_harness_pe_ranked_lists_equal_oracle(g1_db, g1_harness):
    """Pop order of the top-k heap == libstdc++ priority_queue (paired.hpp:51-74)."""
    h = g1_harness[24]
    names, s1, _ = next(refio.load_fastq_batches(os.path.join(refio.GOLDEN, "pe_1.fastq"), 10 ** 7))
    for k in (2, 3, 7, 50, 300):
        for ag, seqs in ((False, s1),):
            r, n, _ = h.pe_topk(seqs, ag, 6, 5000, k)
            ro, no, _ = refio.oracle_pe_topk(g1_db, seqs, ag, 6, 5000, k)
            assert np.array_equal(n, no)
            for j in range(len(seqs)):
                a, b = r[j][:n[j]], ro[j][:no[j]]
                assert np.array_equal(a["genome_pos"], b["genome_pos"]), (k, j)
                assert np.array_equal(a["mismatch"], b["mismatch"]) and np.array_equal(a["strand"], b["strand"])


# ---------------------------------------------------------------------------
# random genomes: many short chromosomes (chromosome-end entries make buckets
# BAD -> literal path), planted repeats, mixed read lengths
# ---------------------------------------------------------------------------
def make_random_case(seed, n_chrom, tmpdir):
    rng = random.Random(seed)
    seqs = []
    unit = "".join(rng.choice("ACGT") for _ in range(220))
    for i in range(n_chrom):
        L = rng.choice([40, 90, 150, 300, 700, 2000, 5000])
        s = [rng.choice("ACGT") for _ in range(L)]
        if L >= 700 and rng.random() < 0.6:  # shared repeat, sometimes running into the chromosome end
            p = rng.randrange(0, L - 100)
            ln = min(220, L - p)
            s[p:p + ln] = unit[:ln]
        seqs.append(("c%d" % i, "".join(s)))
    fa = os.path.join(tmpdir, "rnd_%d.fa" % seed)
    with open(fa, "w") as f:
        for nm, s in seqs:
            f.write(">%s\n%s\n" % (nm, s))
    idx = os.path.join(tmpdir, "rnd_%d.dbindex" % seed)
    assert refio.harness().walt_makedb(fa.encode(), idx.encode(), 4) == 0
    return seqs, refio.DbIndex(idx)


def sample_reads(rng, seqs, n, conv):
    a, b = ("C", "T") if conv == "CT" else ("G", "A")
    out = []
    while len(out) < n:
        nm, g = seqs[rng.randrange(len(seqs))]
        L = rng.choice([38, 45, 60, 100, 100, 100, 131, 134, 135, 150, 200])
        if len(g) < L:
            continue
        p = rng.randrange(0, len(g) - L + 1)
        if rng.random() < 0.3:
            p = rng.choice([0, len(g) - L, max(0, len(g) - L - 1)])  # chromosome edges
        s = g[p:p + L]
        if rng.random() < 0.5:
            s = refio.revcomp(s)
        s = "".join(b if (c == a and rng.random() < 0.9) else c for c in s)
        rate = rng.choice([0.0, 0.01, 0.03])
        s = "".join(rng.choice("ACGT") if rng.random() < rate else c for c in s)
        out.append(s)
    return out


@pytest.mark.parametrize("seed,n_chrom", [(1, 60), (2, 200), (3, 12), (4, 1100)])
def test_harness_random_genomes_vs_oracle(scratch, seed, n_chrom):
    seqs, db = make_random_case(seed, n_chrom, scratch)
    rng = random.Random(seed * 77)
    reads_ct = sample_reads(rng, seqs, 1500, "CT")
    reads_ga = sample_reads(rng, seqs, 700, "GA")
    want_ct, wct = refio.oracle_se(db, reads_ct, ag=False, max_mm=6, b=5000)
    want_ga, _ = refio.oracle_se(db, reads_ga, ag=True, max_mm=4, b=50)
    any_bad = False
    for D in (24, 25, 27):
        h = refio.HarnessIndex(db, D)
        any_bad = any_bad or any(v > 0 for v in h.bad.values())
        got, ts = h.map_se(reads_ct, False, 6, 5000)
        assert_best_equal(got, want_ct, "D=%d CT" % D)
        assert ts == int(wct["too_short"])
        got, _ = h.map_se(reads_ga, True, 4, 50)
        assert_best_equal(got, want_ga, "D=%d GA" % D)
        if D == 25:
            # every bucket through the literal path must give the same answer
            for s in range(4):
                h.force_bad(s, True)
            got, _ = h.map_se(reads_ct, False, 6, 5000)
            assert_best_equal(got, want_ct, "forced literal")
            # paired-end top-k lists against the oracle's priority_queue
            for k in (2, 5, 50):
                r, n, _ = h.pe_topk(reads_ct[:400], False, 6, 5000, k)
                ro, no, _ = refio.oracle_pe_topk(db, reads_ct[:400], False, 6, 5000, k)
                assert np.array_equal(n, no)
                for j in range(400):
                    assert np.array_equal(r[j][:n[j]]["genome_pos"], ro[j][:no[j]]["genome_pos"]), (k, j)
                    assert np.array_equal(r[j][:n[j]]["mismatch"], ro[j][:no[j]]["mismatch"])
        h.close()
    assert any_bad, "test genome should contain BAD (chromosome-end) buckets"


def test_harness_crowded_chromosome_ends(scratch):
    """Many short chromosomes cut from ONE repeated sequence: buckets are crowded with
    chromosome-end entries ('outliers') that share long prefixes with ordinary entries,
    which is where the key search and the literal search could disagree."""
    rng = random.Random(99)
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
    fa = os.path.join(scratch, "crowded.fa")
    with open(fa, "w") as f:
        for nm, s in seqs:
            f.write(">%s\n%s\n" % (nm, s))
    idxp = os.path.join(scratch, "crowded.dbindex")
    assert refio.harness().walt_makedb(fa.encode(), idxp.encode(), 4) == 0
    db = refio.DbIndex(idxp)
    reads = []
    for _ in range(3000):
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
    for D in (24, 28):
        h = refio.HarnessIndex(db, D)
        assert sum(h.bad.values()) > 100
        got, _ = h.map_se(reads, False, 6, 5000)
        assert_best_equal(got, want, "crowded D=%d" % D)
        got, _ = h.map_se(reads, False, 3, 40)
        assert_best_equal(got, want_b, "crowded b=40 D=%d" % D)
        for k in (2, 50):
            r, n, _ = h.pe_topk(reads[:500], False, 6, 5000, k)
            ro, no, _ = refio.oracle_pe_topk(db, reads[:500], False, 6, 5000, k)
            assert np.array_equal(n, no)
            for j in range(500):
                assert np.array_equal(r[j][:n[j]]["genome_pos"], ro[j][:no[j]]["genome_pos"]), (k, j)
        h.close()


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_harness_outlier_stress_low_entropy(scratch, seed):
    """Thousands of chromosome ends in a low-entropy genome (T-rich after conversion, few distinct 12-mers):
    most probes share long prefixes with chromosome-end entries, which is where the refined danger rule
    (core.h probe_is_dangerous: a probe whose character at the outlier's first missing position exceeds the real
    byte there is safe) must agree with the literal search.  Directory search, forced literal search and the
    oracle must give identical records."""
    rng = random.Random(1000 + seed)
    alphabet = "TTTTTTCCAG" if seed % 2 else "TTCCCCAAGG"  # C -> T makes the first one ~80 % T
    motif = "".join(rng.choice(alphabet) for _ in range(400))
    seqs = []
    for i in range(1500):
        L = rng.choice([37, 38, 39, 40, 45, 52, 60, 75, 90, 120, 135])
        a = rng.randrange(0, 400 - 136)
        s = list(motif[a:a + L])
        for _ in range(rng.randrange(0, 3)):
            s[rng.randrange(len(s))] = rng.choice("ACGT")
        seqs.append(("t%d" % i, "".join(s)))
    seqs.append(("long", motif * 4))
    fa = os.path.join(scratch, "stress_%d.fa" % seed)
    with open(fa, "w") as f:
        for nm, s in seqs:
            f.write(">%s\n%s\n" % (nm, s))
    idxp = os.path.join(scratch, "stress_%d.dbindex" % seed)
    assert refio.harness().walt_makedb(fa.encode(), idxp.encode(), 4) == 0
    db = refio.DbIndex(idxp)
    long_seq = motif * 4
    reads = []
    for _ in range(6000):
        L = rng.choice([38, 41, 44, 50, 62, 80, 100, 100, 130])
        a = rng.randrange(0, len(long_seq) - L)
        s = long_seq[a:a + L]
        if rng.random() < 0.5:
            s = refio.revcomp(s)
        s = "".join("T" if (c == "C" and rng.random() < 0.9) else c for c in s)
        s = "".join(rng.choice("ACGT") if rng.random() < 0.02 else c for c in s)
        reads.append(s)
    want, _ = refio.oracle_se(db, reads, max_mm=6, b=5000)
    want_b, _ = refio.oracle_se(db, reads, max_mm=4, b=30)
    for D in (24, 27):
        h = refio.HarnessIndex(db, D)
        # region level: every probe the danger test calls safe gets the literal search's region from the
        # directory/key search -- including the probes that share an outlier's prefix (refined rule)
        probes, differ, dangerous, released = h.region_check(reads)
        assert differ == 0, "%d of %d safe probes differ from the literal search" % (differ, probes)
        assert dangerous > 100 and released > 100, (probes, dangerous, released)
        got, _ = h.map_se(reads, False, 6, 5000)
        assert_best_equal(got, want, "stress D=%d" % D)
        got, _ = h.map_se(reads, False, 4, 30)
        assert_best_equal(got, want_b, "stress b=30 D=%d" % D)
        if D == 24:
            r, n, _ = h.pe_topk(reads[:800], False, 6, 5000, 50)
            ro, no, _ = refio.oracle_pe_topk(db, reads[:800], False, 6, 5000, 50)
            assert np.array_equal(n, no)
            for j in range(800):
                assert np.array_equal(r[j][:n[j]]["genome_pos"], ro[j][:no[j]]["genome_pos"]), j
        h.close()


@pytest.mark.parametrize("n_contigs,genome_bp", [(400, 1_500_000), (3000, 400_000)])
def test_harness_inferred_literal_search(scratch, n_contigs, genome_bp):
    """The literal search of a dangerous probe, inferred (core.h lit_region_inferred, the product's route): the reference's
    bisection is followed mid by mid, but an entry is loaded only where its byte cannot be known from the key search's
    boundaries.  On a low-entropy genome (large buckets) with hundreds / thousands of contig ends -- sparse and crowded
    outliers -- every dangerous probe must get the region of the plain search over the whole bucket (and of the memoised
    and from-the-level routes), with a fraction of its entry reads."""
    rng = random.Random(4242 + n_contigs)
    long_seq = "".join(rng.choice("TTTTTTCCAGAG") for _ in range(genome_bp))
    seqs = [("long", long_seq)]
    for i in range(n_contigs):
        L = rng.choice([60, 90, 150, 300, 500])
        a = rng.randrange(0, len(long_seq) - L)
        s = list(long_seq[a:a + L])
        for _ in range(rng.randrange(0, 4)):
            s[rng.randrange(L)] = rng.choice("ACGT")
        seqs.append(("c%d" % i, "".join(s)))
    fa = os.path.join(scratch, "inf_%d.fa" % n_contigs)
    with open(fa, "w") as f:
        for nm, s in seqs:
            f.write(">%s\n%s\n" % (nm, s))
    idxp = os.path.join(scratch, "inf_%d.dbindex" % n_contigs)
    assert refio.harness().walt_makedb(fa.encode(), idxp.encode(), 4) == 0
    db = refio.DbIndex(idxp)
    reads = []
    for _ in range(3000):
        L = rng.choice([60, 100, 100, 100, 150])
        a = rng.randrange(0, len(long_seq) - L)
        s = long_seq[a:a + L]
        if rng.random() < 0.5:
            s = refio.revcomp(s)
        s = "".join("T" if (c == "C" and rng.random() < 0.95) else c for c in s)
        s = "".join(rng.choice("ACGT") if rng.random() < 0.01 else c for c in s)
        reads.append(s)
    h = refio.HarnessIndex(db, 24)
    refio.HarnessIndex.memo_stats()  # (switches the counters on)
    probes, differ, dangerous, _ = h.region_check(reads)
    searched, inferred, loads, searches, steps, ref_reads = refio.HarnessIndex.memo_stats()[:6]
    assert differ == 0, "%d of %d probes: the routes of the literal search disagree" % (differ, probes)
    assert dangerous > 300 and inferred == dangerous, (dangerous, inferred)
    assert ref_reads > 30 * dangerous, (ref_reads, dangerous)                        # the reference's bisection: tens of reads per probe
    assert loads < 2 * dangerous and searches < 8 * dangerous, (loads, searches, dangerous)  # inferred: a few searches, hardly a load
    want, _ = refio.oracle_se(db, reads, max_mm=6, b=5000)
    got, _ = h.map_se(reads, False, 6, 5000)
    assert_best_equal(got, want, "inferred literal search")
    h.close()


def test_round_based_kary_search_equals_the_two_sided_search():
    """core.h kary_round (heavy stages: both strands' slots advanced in one loop, eight pivots per round, shared by
    the lower- and the upper-bound search while their ranges coincide) must return the equal range that
    slot_kary_search returns: 3,000 random sorted slots of 1..5,000 entries with long runs of equal keys, masks of
    1..32 key characters, six targets each (present and absent), sub-ranges of the slot."""
    for seed in range(3):
        assert refio.harness().hh_kary_check(seed, 1000) == 0



def test_fence_search_equals_the_equal_range():
    """core.h slot_fence_search (round 3: long slots searched through the fence keys -- every 16th, 256th, 4096th,
    65536th entry's key, a static B-tree of fan-out 16 without pointers) must return the equal range of the masked
    key: random sorted arrays of 1..300,000 entries with long runs of equal keys, masks of 1..32 key characters,
    eight targets each (present and absent), the whole array and random sub-ranges as the slot; compared with a
    linear scan."""
    for seed in range(3):
        assert refio.harness().hh_fence_check(seed, 700) == 0


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_harness_tail_characters_narrowed_by_the_verifier(scratch, seed, monkeypatch):
    """DESIGN.md section 4b (round 3): a read above 134 bases has a seed of more than the 44 care characters the entry
    keys hold; the staged kernels hand the whole key-equal range to the verifier when the range lies in dense candidate
    windows, and the verifier keeps the candidates whose characters 44.. equal the read's.  That must be IndexRegion's
    region whenever the range holds no entry with a care character beyond its chromosome's end (the run breakers,
    core.h kTailBreakRoom).  A genome of one sequence holding eight copies of a motif plus a dozen short chromosomes cut
    from the same motif: key-equal ranges are long, and a fifth of them hold chromosome-end entries.  Every deferrable
    range is compared with lit_region (0 differences allowed); and, as a test of the test, with the breaker rule
    switched off differences must show."""
    rng = random.Random(2000 + seed)
    alphabet = "TTTTTTCCAG" if seed % 2 else "ACGT"
    mlen = 2000 + 500 * (seed % 3)
    motif = "".join(rng.choice(alphabet) for _ in range(mlen))
    seqs = []
    for i in range(12 + seed % 3 * 6):
        L = rng.choice([150, 160, 170, 180, 185, 190, 200, 215, 260, 330])
        a = rng.randrange(0, mlen)
        s = list((motif * 3)[a:a + L])
        for _ in range(rng.randrange(0, 3)):
            s[rng.randrange(len(s))] = rng.choice("ACGT")
        seqs.append(("t%d" % i, "".join(s)))
    seqs.append(("long", motif * 8))
    fa = os.path.join(scratch, "tail_%d.fa" % seed)
    with open(fa, "w") as f:
        for nm, s in seqs:
            f.write(">%s\n%s\n" % (nm, s))
    idxp = os.path.join(scratch, "tail_%d.dbindex" % seed)
    assert refio.harness().walt_makedb(fa.encode(), idxp.encode(), 4) == 0
    db = refio.DbIndex(idxp)
    long_seq = motif * 8
    reads = []
    for _ in range(4000):
        L = rng.choice([140, 143, 150, 150, 150, 152])
        a = rng.randrange(0, len(long_seq) - L)
        s = long_seq[a:a + L]
        if rng.random() < 0.5:
            s = refio.revcomp(s)
        s = "".join("T" if (c == "C" and rng.random() < 0.9) else c for c in s)
        s = "".join(rng.choice("ACGT") if rng.random() < 0.01 else c for c in s)
        reads.append(s)
    for D in (24, 28):
        h = refio.HarnessIndex(db, D)
        ranges, deferrable, differ, with_breaker = h.tail_check(reads)
        assert differ == 0, "%d of %d deferrable ranges: the verifier's set differs from IndexRegion's" % (differ, deferrable)
        assert deferrable > 2000 and with_breaker > 400, (ranges, deferrable, with_breaker)
        if D == 24:
            monkeypatch.setenv("WALT_AMD_TEST_DEFER_BREAKERS", "1")
            _, all_ranges, differ_without_rule, _ = h.tail_check(reads)
            monkeypatch.delenv("WALT_AMD_TEST_DEFER_BREAKERS")
            assert all_ranges == ranges and differ_without_rule > 10, (all_ranges, differ_without_rule)
        h.close()


def test_harness_candidate_list_gives_each_lane_its_own_fold():
    """map_se.hip coop_lane_regions replaced "every lane walks its own regions" by one candidate list over the wavefront:
    owner by bisection over the lanes' exclusive counts, one-candidate summaries through a segmented scan with
    summary_merge, the last lane of a run handed back to its owner turn after turn.  The harness runs exactly those steps
    on 64 simulated lanes (regions of up to 4 / 16 / 64 slots on both strands, lanes without regions, ties in mismatch
    count and position) and compares every lane's two summaries with the in-order fold of its own candidates."""
    assert refio.harness().hh_candidate_list_check(20261005, 3000) == 0
