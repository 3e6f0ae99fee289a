#!/usr/bin/env python3
"""Randomised parity soak on the GPU box: fresh random genomes (many short chromosomes, planted repeats,
low-entropy stretches) and read sets with random options, mapped through the C ABI and compared record by
record with the oracle (tests/refio.py; test infrastructure).  Runs until --seconds are used up.

  python3 tools/soak.py --seconds 300 [--seed0 1] [--pattern 3|5|7]
Prints one summary line; exits non-zero at the first difference (with the offending case)."""
import argparse
import os
import random
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def make_genome(rng, pattern):
    n_chrom = rng.choice([3, 12, 60, 250, 900])
    low = rng.random() < 0.4
    alphabet = rng.choice(["TTTTTCCAG", "TTCCCAAGG"]) if low else "ACGT"
    unit = "".join(rng.choice(alphabet) for _ in range(300))
    seqs = []
    for i in range(n_chrom):
        L = rng.choice([36, 37, 38, 40, 52, 90, 150, 300, 700, 2000, 6000])
        s = [rng.choice(alphabet) for _ in range(L)]
        if L >= 150 and rng.random() < 0.6:
            p = rng.randrange(0, L - 60)
            ln = min(len(unit), L - p)
            s[p:p + ln] = unit[:ln]
        seqs.append(("c%d" % i, "".join(s)))
    seqs.append(("big", "".join(rng.choice(alphabet) for _ in range(20000)) + unit * 3))
    if rng.random() < 0.12:  # now and then a few Mbp, so that buckets and directory slots fill up
        big = np.random.default_rng(rng.randrange(1 << 30)).integers(0, len(alphabet), rng.choice([1500000, 4000000]))
        seqs.append(("huge", "".join(np.array(list(alphabet))[big])))
    return seqs


def sample(rng, seqs, n, conv, lengths, refio):
    a, b = ("C", "T") if conv == "CT" else ("G", "A")
    out = []
    while len(out) < n:
        _, g = seqs[rng.randrange(len(seqs))]
        L = rng.choice(lengths)
        if len(g) < L:
            continue
        p = rng.randrange(0, len(g) - L + 1)
        if rng.random() < 0.3:
            p = rng.choice([0, len(g) - L, max(0, len(g) - L - 1)])
        s = g[p:p + L]
        if rng.random() < 0.5:
            s = refio.revcomp(s)
        s = "".join(b if (c == a and rng.random() < 0.9) else c for c in s)
        rate = rng.choice([0.0, 0.01, 0.04])
        out.append("".join(rng.choice("ACGT") if rng.random() < rate else c for c in s))
    return out


class SoakMismatch(AssertionError):
    pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed0", type=int, default=1)
    ap.add_argument("--pattern", type=int, default=3, choices=[3, 5, 7])
    ap.add_argument("--genomes", type=int, default=0, help="stop after this many genomes (0: when --seconds are used up)")
    args = ap.parse_args()
    try:
        print(run_soak(args.seconds, args.seed0, args.pattern, args.genomes or None))
    except SoakMismatch as e:
        print(e)
        sys.exit(1)


def run_soak(seconds, seed0=1, pattern=3, max_genomes=None):
    """Genomes seed0, seed0 + 1, ... until `seconds` are used up or `max_genomes` are done; returns the summary line,
    raises SoakMismatch at the first difference (tests/test_gpu_soak.py runs a fixed set of seeds this way)."""
    from types import SimpleNamespace
    args = SimpleNamespace(seconds=seconds, seed0=seed0, pattern=pattern)
    import refio
    import walt_amd
    refio.set_pattern(args.pattern)
    walt_amd.set_pattern(args.pattern)
    lo, hi = refio.MIN_READ_LEN[args.pattern], min(refio.MAX_READ_LEN[args.pattern], 260)
    t_end = time.time() + args.seconds
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    cases = reads_total = pairs_total = 0
    seed = args.seed0
    t_note = time.time()
    while time.time() < t_end and (max_genomes is None or cases < max_genomes):
        if time.time() - t_note > 60:  # a sign of life (runs under a watchdog that takes minutes of silence for a hang)
            print("soak: %d genomes so far, no difference" % cases, file=sys.stderr, flush=True)
            t_note = time.time()
        rng = random.Random(seed)
        tmp = tempfile.mkdtemp(prefix="walt_soak_", dir=base)
        try:
            seqs = make_genome(rng, args.pattern)
            fa = os.path.join(tmp, "g.fa")
            with open(fa, "w") as f:
                for nm, s in seqs:
                    f.write(">%s\n%s\n" % (nm, s))
            path = os.path.join(tmp, "g.dbindex")
            walt_amd.makedb(fa, path, threads=4)
            db = refio.DbIndex(path)
            D = rng.choice([-1, 24, 26, 29])
            if rng.random() < 0.3:
                os.environ["WALT_AMD_TABLE"] = "1"
            else:
                os.environ.pop("WALT_AMD_TABLE", None)
            idx = walt_amd.Index.open(path, device=0, dir_bits=D)
            # a third of the genomes on one or two blocks: every wavefront of the staged kernels then walks through several
            # windows of its list (the read hand-out of k_se_stage), which 1,500 reads on the full grid never do
            g_opt = rng.choice([0, 0, 0, 0, 1, 2])
            if g_opt:
                idx.set_option("grid", g_opt)
            # a quarter of the genomes with the paired-end literal round seed by seed (default: one launch when the list
            # is short, map_pe.hip k_pe_stage); drawn from a generator of its own so that a seed's genome and reads stay
            if random.Random(seed * 7919 + 13).random() < 0.25:
                idx.set_option("pe_lit_fuse", 0)
            lengths = [lo + 2, lo + 3, 40, 45, 60, 100, 100, 100, 131, 140, min(150, hi), min(200, hi), hi]
            # the kernels are instantiated per read-length class (up to 112, 128, 160 ... bases: the batch's longest read
            # selects the instance): some genomes get batches that stop at 112 or 128 bases
            cls = rng.choice(["all", "all", "le112", "le128"])
            if cls == "le112":
                lengths = [x for x in lengths if x <= 112] + [90, 96, 104, 110, 112]
            elif cls == "le128":
                lengths = [x for x in lengths if x <= 128] + [100, 113, 119, 120, 125, 128]
            m, b, k = rng.choice([0, 2, 6, 10]), rng.choice([2, 30, 5000]), rng.choice([2, 5, 50, 300])
            for conv, ag in (("CT", False), ("GA", True)):
                reads = sample(rng, seqs, 1500, conv, lengths, refio)
                want, _ = refio.oracle_se(db, reads, ag=ag, max_mm=m, b=b)
                got, _ = idx.map_se_batch(*walt_amd.pack_reads(reads), ag_wildcard=ag, max_mismatches=m, b=b)
                for f in ("genome_pos", "times", "strand", "mismatch"):
                    if not np.array_equal(got[f], want[f]):
                        bad = int(np.nonzero(got[f] != want[f])[0][0])
                        raise SoakMismatch("MISMATCH seed %d %s field %s read %d (%s) D=%d m=%d b=%d" % (seed, conv, f, bad, reads[bad], D, m, b))
                reads_total += len(reads)
            s1 = sample(rng, seqs, 500, "CT", lengths, refio)
            s2 = sample(rng, seqs, 500, "GA", lengths, refio)
            L = rng.choice([200, 1000])
            res, _ = idx.map_pe_batch(*walt_amd.pack_reads(s1), *walt_amd.pack_reads(s2), max_mismatches=m, b=b, top_k=k,
                                      frag_range=L)
            wantp, _, _ = refio.oracle_pe(db, s1, s2, max_mm=m, b=b, top_k=k, frag_range=L)
            for f in ("best_times", "frag_len", "pair_mm", "best_i", "best_j"):
                if not np.array_equal(res[f], wantp[f]):
                    bad = int(np.nonzero(res[f] != wantp[f])[0][0])
                    raise SoakMismatch("MISMATCH seed %d paired-end field %s pair %d D=%d m=%d b=%d k=%d L=%d" % (seed, f, bad, D, m, b, k, L))
            for mate in ("m1", "m2"):
                for f in ("genome_pos", "times", "strand", "mismatch"):
                    if not np.array_equal(res[mate][f], wantp[mate][f]):
                        raise SoakMismatch("MISMATCH seed %d paired-end %s.%s D=%d m=%d b=%d k=%d L=%d" % (seed, mate, f, D, m, b, k, L))
            pairs_total += len(s1)
            idx.close()
            cases += 1
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        seed += 1
    os.environ.pop("WALT_AMD_TABLE", None)
    return "soak ok: pattern %d, %d genomes, %d single-end reads and %d pairs identical to the oracle (seeds %d..%d)" % (
        args.pattern, cases, reads_total, pairs_total, args.seed0, seed - 1)


if __name__ == "__main__":
    main()
