"""Round-3 GPU tests (run with -m gpu on an MI355X): reads above 134 bases whose seeds outrun the 44 care characters
of the entry keys -- the staged kernels leave the narrowing by the characters behind the key to the verifier
 -- on a genome whose key-equal ranges are long and full of chromosome ends."""
import os
import random

import numpy as np
import pytest

import refio
from test_harness_cpu import assert_best_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wa():
    import walt_amd
    assert walt_amd.device_count() >= 1, "no HIP device: the walt_amd hot path has no CPU fallback"
    return walt_amd


@pytest.fixture(scope="module")
def tail_case(scratch):
    """One sequence holding 40 copies of a 2,500-base motif (2 % of the copies' bases changed, so that key-equal ranges
    split on the characters behind the key), 30 short chromosomes cut from the motif (chromosome-end entries inside the
    ranges: run breakers), and 2,000 exact copies of a 170-base unit (ranges the samples prove larger than -b)."""
    rng = random.Random(4242)
    motif = "".join(rng.choice("ACGT") for _ in range(2500))
    unit = "".join(rng.choice("ACGT") for _ in range(170))

    def mutate(s, d):
        return "".join(rng.choice("ACGT") if rng.random() < d else c for c in s)

    seqs = []
    for i in range(30):
        L = rng.choice([150, 160, 170, 180, 185, 190, 200, 215, 260, 330])
        a = rng.randrange(0, 2500)
        seqs.append(("t%d" % i, mutate((motif * 2)[a:a + L], 0.01)))
    seqs.append(("long", "".join(mutate(motif, 0.02) for _ in range(40))))
    seqs.append(("sat", unit * 2000))
    fa = os.path.join(scratch, "tail_gpu.fa")
    with open(fa, "w") as f:
        for nm, s in seqs:
            f.write(">%s\n%s\n" % (nm, s))
    import walt_amd
    idxp = os.path.join(scratch, "tail_gpu.dbindex")
    walt_amd.makedb(fa, idxp, threads=4)
    db = refio.DbIndex(idxp)
    long_seq = motif * 3
    sat_seq = unit * 4

    def sample(n, conv, lengths):
        a_, b_ = ("C", "T") if conv == "CT" else ("G", "A")
        out = []
        for _ in range(n):
            L = rng.choice(lengths)
            src = sat_seq if rng.random() < 0.1 else long_seq
            p = rng.randrange(0, len(src) - L)
            s = src[p:p + L]
            if rng.random() < 0.5:
                s = refio.revcomp(s)
            s = "".join(b_ if (c == a_ and rng.random() < 0.9) else c for c in s)
            out.append(mutate(s, rng.choice([0.0, 0.01, 0.03])))
        return out
    return db, idxp, sample


@pytest.mark.parametrize("b", [5000, 300, 20])
def test_gpu_long_seeds_narrowed_by_the_verifier_single_end(wa, tail_case, b):
    db, idxp, sample = tail_case
    idx = wa.Index.open(idxp, device=0)
    assert idx.window_entries(0) > 50000, "the case must have dense candidate windows (the deferral only happens inside them)"
    for conv, ag in (("CT", False), ("GA", True)):
        reads = sample(3000, conv, [135, 140, 143, 146, 150, 150, 150, 152, 158, 160])
        want, _ = refio.oracle_se(db, reads, ag=ag, max_mm=10, b=b)
        got, _ = idx.map_se_batch(*wa.pack_reads(reads), ag_wildcard=ag, max_mismatches=10, b=b)
        assert_best_equal(got, want, "tail %s b=%d" % (conv, b))
        assert int((want["times"] >= 1).sum()) > 1500
    idx.close()


@pytest.mark.parametrize("top_k,b", [(50, 5000), (3, 300), (300, 20)])
def test_gpu_long_seeds_narrowed_by_the_verifier_paired_end(wa, tail_case, top_k, b):
    db, idxp, sample = tail_case
    idx = wa.Index.open(idxp, device=0)
    s1 = sample(1500, "CT", [140, 150, 150, 152, 160])
    s2 = sample(1500, "GA", [140, 150, 150, 152, 160])
    res, _ = idx.map_pe_batch(*wa.pack_reads(s1), *wa.pack_reads(s2), max_mismatches=10, b=b, top_k=top_k, frag_range=1000)
    want, _, _ = refio.oracle_pe(db, s1, s2, max_mm=10, b=b, top_k=top_k, frag_range=1000)
    for f in ("best_times", "frag_len", "pair_mm", "best_i", "best_j"):
        assert np.array_equal(res[f], want[f]), f
    for mate in ("m1", "m2"):
        for f in ("genome_pos", "times", "strand", "mismatch"):
            assert np.array_equal(res[mate][f], want[mate][f]), (mate, f)
    idx.close()


def test_gpu_ranges_larger_than_b_are_skipped_like_the_reference(wa, tail_case):
    """A key-equal range whose samples already prove more than -b members is skipped without being streamed
    (map_items.h tail_items_narrow); one just below -b is streamed and counted.  Reads from the exact 170-base array
    (2,000 copies) and the 40-copy motif at three values of -b around those sizes, against the oracle."""
    db, idxp, sample = tail_case
    idx = wa.Index.open(idxp, device=0)
    unit_reads = sample(4000, "CT", [150, 152, 160])
    for b in (5000, 1500, 100):
        want, _ = refio.oracle_se(db, unit_reads, max_mm=10, b=b)
        got, _ = idx.map_se_batch(*wa.pack_reads(unit_reads), max_mismatches=10, b=b)
        assert_best_equal(got, want, "array b=%d" % b)
    idx.close()
