#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference.

Run in the build container only (needs oracle/_ref/{makedb,walt}, built by
`make -f oracle/Makefile.ref` from /root/reference).  Commits DATA only:
  g1.fa                     N-free genome (deterministic makedb)
  *.fastq                   read sets (SE C->T, SE A/G, PE, edge cases)
  cases.json                every reference invocation (args) + md5 of index files
  out/<case>/*.gz           the reference's outputs (SAM / MR / mapstats / _ambiguous / _unmapped)
Nothing from the reference source tree is stored.
"""
import gzip
import hashlib
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_BIN = os.path.join(ROOT, "oracle", "_ref")

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def make_genome(rng):
    """4 sequences with planted repeats; N-free so makedb is deterministic."""
    chrA = list(rand_seq(rng, 60000))
    chrB = list(rand_seq(rng, 30000))
    chrC = list(rand_seq(rng, 8000))
    chrD = list(rand_seq(rng, 500))
    # 300 bp segment: 3x in chrA, 1x in chrB (SURVEY 8c second probe set)
    seg = rand_seq(rng, 300)
    for off in (5000, 21000, 47000):
        chrA[off:off + 300] = seg
    chrB[7050:7350] = seg
    # 80 copies of a 160 bp unit separated by 40 bp unique spacers in chrB
    # (more copies than top_k=50 -> heap eviction order)
    unit = rand_seq(rng, 160)
    p = 10000
    for _ in range(80):
        chrB[p:p + 160] = unit
        p += 200
    # 12 near-copies (2 substitutions each) of a 200 bp unit in chrC
    unit2 = rand_seq(rng, 200)
    p = 1000
    for _ in range(12):
        u = list(unit2)
        for _ in range(2):
            k = rng.randrange(200)
            u[k] = rng.choice([c for c in "ACGT" if c != u[k]])
        chrC[p:p + 200] = u
        p += 450
    # a low-complexity stretch and a poly-T run in chrA
    chrA[30000:30400] = list("ACT" * 133 + "A")
    chrA[33000:33300] = list("T" * 300)
    # lower-case some bases (makedb upper-cases, reference.cpp:122)
    for i in range(1000, 1400):
        chrA[i] = chrA[i].lower()
    return [("chrA first sequence", "".join(chrA)), ("chrB\tsecond", "".join(chrB)),
            ("chrC", "".join(chrC)), ("chrD_short", "".join(chrD))]


def write_fasta(path, seqs):
    with open(path, "w") as f:
        for name, s in seqs:
            f.write(">" + name + "\n")
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + "\n")


def bisulfite(rng, s, conv):
    """conv='CT': C->T at 95% of Cs; conv='GA': G->A at 95% of Gs."""
    a, b = ("C", "T") if conv == "CT" else ("G", "A")
    return "".join((b if (c == a and rng.random() < 0.95) else c) for c in s)


def mutate(rng, s, rate):
    out = list(s)
    for i in range(len(out)):
        if rng.random() < rate:
            out[i] = rng.choice([c for c in "ACGT" if c != out[i]])
    return "".join(out)


def qual(rng, n):
    return "".join(chr(33 + rng.randrange(2, 41)) for _ in range(n))


def sample_read(rng, seqs, L, conv, rate, chrom=None, pos=None, strand=None):
    ci = rng.randrange(len(seqs) - 1) if chrom is None else chrom  # skip tiny chrD by default
    g = seqs[ci][1].upper()
    if pos is None:
        pos = rng.randrange(0, len(g) - L + 1)
    if strand is None:
        strand = rng.choice("+-")
    s = g[pos:pos + L]
    if strand == "-":
        s = revcomp(s)
    s = mutate(rng, bisulfite(rng, s, conv), rate)
    return s


def write_fastq(path, recs):
    with open(path, "w") as f:
        for name, s, q in recs:
            f.write("@%s\n%s\n+\n%s\n" % (name, s, q))


def make_se_reads(rng, seqs, conv, n, tag):
    recs = []
    A = seqs[0][1].upper()
    B = seqs[1][1].upper()

    def add(name, s):
        recs.append(("%s_%d_%s" % (tag, len(recs), name), s, qual(rng, len(s))))

    # --- known-answer probes on a unique 100/150 bp window of chrA
    def probe(L, mm_offsets, pos=12345):
        s = list(bisulfite(rng, A[pos:pos + L], conv))
        for o in mm_offsets:
            # substitute with a base that stays a mismatch after conversion
            cands = [c for c in "ACGT" if c != s[o]]
            if conv == "CT":
                cands = [c for c in cands if not ({c, s[o]} == {"C", "T"})]
            else:
                cands = [c for c in cands if not ({c, s[o]} == {"G", "A"})]
            s[o] = cands[0]
        return "".join(s)

    for name, L, offs in [("exact", 100, []), ("mm_1_2", 100, [1, 2]), ("mm_1_2_70", 100, [1, 2, 70]),
                          ("mm_1_2_71", 100, [1, 2, 71]), ("mm_1_70", 100, [1, 70]),
                          ("x150_exact", 150, []), ("x150_1_2_142", 150, [1, 2, 142]),
                          ("x150_1_2_143", 150, [1, 2, 143]), ("x150_1_2_70_142", 150, [1, 2, 70, 142]),
                          ("mm7", 100, [3, 9, 20, 33, 47, 62, 90]), ("mm6", 100, [3, 9, 20, 33, 47, 62])]:
        add(name, probe(L, offs))
    # --- chromosome edges: offset 0, ending exactly at the end, one base earlier (mapping.cpp:285)
    for ci in (0, 1, 2):
        g = seqs[ci][1].upper()
        add("edge_start_c%d" % ci, bisulfite(rng, g[0:100], conv))
        add("edge_end_c%d" % ci, bisulfite(rng, g[len(g) - 100:], conv))
        add("edge_end1_c%d" % ci, bisulfite(rng, g[len(g) - 101:len(g) - 1], conv))
        add("edge_rc_start_c%d" % ci, bisulfite(rng, revcomp(g[len(g) - 100:]), conv))
        add("edge_rc_end_c%d" % ci, bisulfite(rng, revcomp(g[0:100]), conv))
        add("edge_rc_end1_c%d" % ci, bisulfite(rng, revcomp(g[1:101]), conv))
    # --- repeats: 4-copy segment, 80-copy unit, near-copies
    add("rep4", bisulfite(rng, A[5100:5200], conv))
    add("rep4_mm", mutate(rng, bisulfite(rng, A[5100:5200], conv), 0.02))
    add("rep4_rc", bisulfite(rng, revcomp(A[5100:5200]), conv))
    add("rep80", bisulfite(rng, B[10010:10110], conv))
    add("rep80_rc", bisulfite(rng, revcomp(B[10010:10110]), conv))
    add("rep80_span", bisulfite(rng, B[10100:10200], conv))
    for k in range(6):
        add("near%d" % k, sample_read(rng, seqs, 100, conv, 0.0, chrom=2, pos=1000 + 450 * k + 20, strand="+"))
    add("lowcomplex", bisulfite(rng, A[30050:30150], conv))
    add("polyT", A[33050:33150])
    # --- short / boundary lengths (MINIMALREADLEN 38)
    for L in (20, 37, 38, 39, 40, 41, 50, 59, 60, 61, 75, 101, 125, 149, 150, 151, 152, 153, 180, 250):
        add("len%d" % L, sample_read(rng, seqs, L, conv, 0.005))
    # --- N and lower-case bases (toACGT RNG, util.hpp:156-163; depends on -N batching)
    for k in range(12):
        s = list(sample_read(rng, seqs, 100, conv, 0.0))
        for _ in range(rng.randrange(1, 4)):
            s[rng.randrange(100)] = "N"
        add("withN%d" % k, "".join(s))
    for k in range(4):
        s = list(sample_read(rng, seqs, 100, conv, 0.0))
        for i in range(40, 44):
            s[i] = s[i].lower()
        add("lower%d" % k, "".join(s))
    # --- tiny chromosome chrD
    D = seqs[3][1].upper()
    add("chrD_a", bisulfite(rng, D[100:200], conv))
    add("chrD_b", bisulfite(rng, revcomp(D[350:450]), conv))
    # --- bulk random reads, 1% substitutions, a few with heavier damage
    while len(recs) < n:
        r = rng.random()
        rate = 0.01 if r < 0.85 else (0.04 if r < 0.95 else 0.10)
        L = 100 if rng.random() < 0.8 else rng.choice([50, 76, 120, 150])
        add("r", sample_read(rng, seqs, L, conv, rate))
    # names with a space (cut at first space, mapping.cpp:87-95)
    recs[5] = (recs[5][0] + " extra words", recs[5][1], recs[5][2])
    return recs


def make_pe_reads(rng, seqs, n, tag):
    r1, r2 = [], []

    def add(name, frag_top, L1=100, L2=100, rate=0.01):
        # fragment of the bisulfite-converted top or bottom strand
        m1 = mutate(rng, frag_top[:L1], rate)
        m2 = mutate(rng, revcomp(frag_top)[:L2], rate)
        nm = "%s_%d_%s" % (tag, len(r1), name)
        r1.append((nm + "/1", m1, qual(rng, len(m1))))
        r2.append((nm + "/2", m2, qual(rng, len(m2))))

    def frag(ci, pos, flen, strand):
        g = seqs[ci][1].upper()[pos:pos + flen]
        if strand == "-":
            g = revcomp(g)
        return bisulfite(rng, g, "CT")

    # specific: repeats (top-k), overlapping mates, tiny fragments, too-long fragments, different chroms
    add("rep80_pair", frag(1, 10010, 300, "+"))
    add("rep80_pair_rc", frag(1, 10410, 300, "-"))
    add("rep4_pair", frag(0, 5050, 260, "+"))
    add("overlap", frag(0, 40000, 130, "+"))
    add("full_overlap", frag(0, 41000, 100, "-"))
    add("frag_1001", frag(0, 15000, 1001, "+"))
    add("frag_1000", frag(0, 16000, 1000, "+"))
    add("frag_999_rc", frag(0, 17000, 999, "-"))
    f1 = frag(0, 2000, 300, "+")
    f2 = frag(1, 2000, 300, "+")
    add("chimera", f1[:150] + f2[150:])
    add("short_mate", frag(0, 3000, 300, "+"), L1=30, L2=100)
    add("len150", frag(0, 52000, 400, "+"), L1=150, L2=150)
    add("len_mixed", frag(2, 4000, 320, "-"), L1=75, L2=125)
    for k in range(6):
        add("near%d" % k, frag(2, 1000 + 450 * k, 380, "+"))
    for k in range(8):
        f = list(frag(rng.randrange(3), 500 + 97 * k, 250, rng.choice("+-")))
        f[10] = "N"
        f[-7] = "N"
        add("withN%d" % k, "".join(f))
    while len(r1) < n:
        ci = rng.randrange(3)
        glen = len(seqs[ci][1])
        flen = rng.randrange(120, 501)
        pos = rng.randrange(0, glen - flen + 1)
        r = rng.random()
        rate = 0.01 if r < 0.85 else (0.04 if r < 0.95 else 0.10)
        add("p", frag(ci, pos, flen, rng.choice("+-")), rate=rate)
    return r1, r2


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def run(cmd, cwd):
    subprocess.run(cmd, cwd=cwd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def main():
    if not os.path.exists(os.path.join(REF_BIN, "walt")):
        sys.exit("build oracle/_ref first: make -f oracle/Makefile.ref")
    rng = random.Random(20261003)
    seqs = make_genome(rng)
    write_fasta(os.path.join(HERE, "g1.fa"), seqs)
    write_fastq(os.path.join(HERE, "se_ct.fastq"), make_se_reads(rng, seqs, "CT", 1200, "sct"))
    write_fastq(os.path.join(HERE, "se_ga.fastq"), make_se_reads(rng, seqs, "GA", 600, "sga"))
    p1, p2 = make_pe_reads(rng, seqs, 800, "pe")
    write_fastq(os.path.join(HERE, "pe_1.fastq"), p1)
    write_fastq(os.path.join(HERE, "pe_2.fastq"), p2)

    # adaptor-contaminated sets for -C (util.hpp:189-233): separate RNG so the sets above stay unchanged
    rng2 = random.Random(777)
    AD1, AD2 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"

    def contaminate(recs, adaptor):
        out = []
        for nm, sq, q in recs:
            r = rng2.random()
            if len(sq) >= 60 and r < 0.5:
                keep = rng2.choice([len(sq) - 3, len(sq) - 5, len(sq) - 6, len(sq) - 9, len(sq) - 14, len(sq) - 20, len(sq) - 33])
                tail = adaptor[:len(sq) - keep]
                if rng2.random() < 0.3 and len(tail) > 6:  # one error inside the adaptor part
                    k = rng2.randrange(len(tail))
                    tail = tail[:k] + rng2.choice([c for c in "ACGT" if c != tail[k]]) + tail[k + 1:]
                sq = sq[:keep] + tail
            out.append((nm, sq, q))
        return out

    se_all = make_se_reads(random.Random(4242), seqs, "CT", 500, "clip")
    write_fastq(os.path.join(HERE, "se_clip.fastq"), contaminate(se_all, AD1))
    c1, c2 = make_pe_reads(random.Random(4243), seqs, 400, "pclip")
    write_fastq(os.path.join(HERE, "pe_clip_1.fastq"), contaminate(c1, AD1))
    write_fastq(os.path.join(HERE, "pe_clip_2.fastq"), contaminate(c2, AD2))

    tmp = tempfile.mkdtemp(prefix="walt_golden_")
    idx = os.path.join(tmp, "g1.dbindex")
    run([os.path.join(REF_BIN, "makedb"), "-c", os.path.join(HERE, "g1.fa"), "-o", idx], tmp)
    index_md5 = {sfx: md5(idx + sfx) for sfx in ("", "_CT00", "_CT01", "_GA10", "_GA11")}

    cases = {
        # name: (kind, extra args)
        "se_mr": ("se_ct", []),
        "se_sam": ("se_ct", ["-sam"]),
        "se_sam_au": ("se_ct", ["-sam", "-a", "-u"]),
        "se_mr_au": ("se_ct", ["-a", "-u"]),
        "se_sam_au_m10": ("se_ct", ["-sam", "-a", "-u", "-m", "10"]),
        "se_sam_au_m2": ("se_ct", ["-sam", "-a", "-u", "-m", "2"]),
        "se_sam_au_b2": ("se_ct", ["-sam", "-a", "-u", "-b", "2"]),
        "se_sam_au_N100": ("se_ct", ["-sam", "-a", "-u", "-N", "100"]),
        "se_sam_au_t4": ("se_ct", ["-sam", "-a", "-u", "-t", "4"]),
        "se_ag_mr_au": ("se_ga", ["-A", "-a", "-u"]),
        "se_ag_sam_au": ("se_ga", ["-A", "-sam", "-a", "-u"]),
        "pe_mr": ("pe", []),
        "pe_sam": ("pe", ["-sam"]),
        "pe_sam_au": ("pe", ["-sam", "-a", "-u"]),
        "pe_mr_au": ("pe", ["-a", "-u"]),
        "pe_sam_au_k3": ("pe", ["-sam", "-a", "-u", "-k", "3"]),
        "pe_sam_au_k300": ("pe", ["-sam", "-a", "-u", "-k", "300"]),
        "pe_sam_au_L200": ("pe", ["-sam", "-a", "-u", "-L", "200"]),
        "pe_sam_au_m2": ("pe", ["-sam", "-a", "-u", "-m", "2"]),
        "pe_sam_au_m10_b20": ("pe", ["-sam", "-a", "-u", "-m", "10", "-b", "20"]),
        "pe_sam_au_N250": ("pe", ["-sam", "-a", "-u", "-N", "250"]),
        "se_clip_sam_au": ("se_clip", ["-sam", "-a", "-u", "-C", AD1]),
        "se_clip_mr_au_N100": ("se_clip", ["-a", "-u", "-C", AD1[:20], "-N", "100"]),
        "pe_clip_sam_au": ("pe_clip", ["-sam", "-a", "-u", "-C", AD1 + ":" + AD2]),
        "pe_clip_mr_au": ("pe_clip", ["-a", "-u", "-C", AD1[:25]]),
    }
    outroot = os.path.join(HERE, "out")
    shutil.rmtree(outroot, ignore_errors=True)
    meta = {"index_md5": index_md5, "cases": {}}
    for name, (kind, extra) in cases.items():
        wd = os.path.join(tmp, name)
        os.makedirs(wd)
        out = os.path.join(wd, "out.sam" if "-sam" in extra else "out.mr")
        cmd = [os.path.join(REF_BIN, "walt"), "-i", idx, "-o", out] + extra
        if kind in ("pe", "pe_clip"):
            cmd += ["-1", os.path.join(HERE, kind + "_1.fastq"), "-2", os.path.join(HERE, kind + "_2.fastq")]
        else:
            cmd += ["-r", os.path.join(HERE, kind + ".fastq")]
        run(cmd, wd)
        dst = os.path.join(outroot, name)
        os.makedirs(dst)
        files = sorted(os.listdir(wd))
        for fn in files:
            with open(os.path.join(wd, fn), "rb") as fi, open(os.path.join(dst, fn + ".gz"), "wb") as fo:
                with gzip.GzipFile(fileobj=fo, mode="wb", mtime=0) as gz:
                    gz.write(fi.read())
        meta["cases"][name] = {"kind": kind, "args": extra, "files": files}
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp)
    print("golden written:", len(cases), "cases")


if __name__ == "__main__":
    main()
