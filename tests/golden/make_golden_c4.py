#!/usr/bin/env python3
"""Golden fixtures for BASELINE.json configs[4] (150 bp reads, -m 10, the A/G-wildcard path and the read length
at which the second table typo of seedpattern.hpp:454 becomes reachable), ADDED to tests/golden/cases.json and
tests/golden/out/ by running the REAL reference binary (oracle/_ref/walt, built by oracle/Makefile.ref from
/root/reference).  Build container only.  Data only is committed:
  se150_ga.fastq            500 A-rich (G->A) single-end reads of 150 bases on g1.fa
  pe150_1.fastq / _2.fastq  500 pairs 2 x 150 bases (mate 1 C->T, mate 2 G->A), fragments of 150..500 bases
  cases:  se150_ag_sam_au_m10, se150_ag_mr_au_m10  (-A -m 10)      pe150_sam_au_m10, pe150_mr_au_m10  (-m 10)
The -P (PBAT) form of the paired cases has no reference implementation in this snapshot; its expected
files are the mate-exchanged rewrite of these outputs (tests/test_gpu_cli.py)."""
import gzip
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (read samplers of the main fixture set)


def read_fasta(path):
    seqs, name, buf = [], None, []
    for ln in open(path):
        ln = ln.rstrip("\n")
        if ln.startswith(">"):
            if name is not None:
                seqs.append((name, "".join(buf)))
            name, buf = ln[1:], []
        else:
            buf.append(ln)
    seqs.append((name, "".join(buf)))
    return seqs


def main():
    ref_walt, ref_makedb = os.path.join(mg.REF_BIN, "walt"), os.path.join(mg.REF_BIN, "makedb")
    if not os.path.exists(ref_walt):
        sys.exit("build oracle/_ref first: make -f oracle/Makefile.ref")
    seqs = read_fasta(os.path.join(HERE, "g1.fa"))
    rng = random.Random(150150)
    A = seqs[0][1].upper()
    B = seqs[1][1].upper()
    se = []

    def add(name, s):
        se.append(("c4_%d_%s" % (len(se), name), s, mg.qual(rng, len(s))))

    def probe(offs, pos=23456):  # known answers around the table typos: offsets 70 and 142 are never counted via seed 2
        s = list(mg.bisulfite(rng, A[pos:pos + 150], "GA"))
        for o in offs:
            cands = [c for c in "ACGT" if c != s[o] and {c, s[o]} != {"G", "A"}]
            s[o] = cands[0]
        return "".join(s)

    for name, offs in (("exact", []), ("mm_1_2_142", [1, 2, 142]), ("mm_1_2_143", [1, 2, 143]), ("mm_1_2_70_142", [1, 2, 70, 142]),
                       ("mm_1_70", [1, 70]), ("mm_1_142", [1, 142]), ("mm10", [3, 9, 20, 33, 47, 62, 90, 105, 120, 139]),
                       ("mm11", [3, 9, 20, 33, 47, 62, 90, 105, 120, 139, 146])):
        add(name, probe(offs))
    for ci in (0, 1, 2):  # chromosome edges with 150-base reads
        g = seqs[ci][1].upper()
        add("edge_start_c%d" % ci, mg.bisulfite(rng, g[0:150], "GA"))
        add("edge_end_c%d" % ci, mg.bisulfite(rng, g[len(g) - 150:], "GA"))
        add("edge_end1_c%d" % ci, mg.bisulfite(rng, g[len(g) - 151:len(g) - 1], "GA"))
        add("edge_rc_end1_c%d" % ci, mg.bisulfite(rng, mg.revcomp(g[1:151]), "GA"))
    add("rep4", mg.bisulfite(rng, A[5100:5250], "GA"))
    add("rep80", mg.bisulfite(rng, B[10010:10160], "GA"))
    add("rep80_rc", mg.bisulfite(rng, mg.revcomp(B[10210:10360]), "GA"))
    for k in range(6):
        add("near%d" % k, mg.sample_read(rng, seqs, 150, "GA", 0.0, chrom=2, pos=1000 + 450 * k + 20, strand="+"))
    for k in range(8):
        s = list(mg.sample_read(rng, seqs, 150, "GA", 0.0))
        for _ in range(rng.randrange(1, 4)):
            s[rng.randrange(150)] = "N"
        add("withN%d" % k, "".join(s))
    while len(se) < 500:
        r = rng.random()
        rate = 0.01 if r < 0.8 else (0.04 if r < 0.93 else 0.09)
        add("r", mg.sample_read(rng, seqs, 150, "GA", rate))
    mg.write_fastq(os.path.join(HERE, "se150_ga.fastq"), se)

    r1, r2 = [], []

    def addp(name, frag_top, rate=0.01):
        m1 = mg.mutate(rng, frag_top[:150], rate)
        m2 = mg.mutate(rng, mg.revcomp(frag_top)[:150], rate)
        nm = "c4p_%d_%s" % (len(r1), name)
        r1.append((nm + "/1", m1, mg.qual(rng, len(m1))))
        r2.append((nm + "/2", m2, mg.qual(rng, len(m2))))

    def frag(ci, pos, flen, strand):
        g = seqs[ci][1].upper()[pos:pos + flen]
        if strand == "-":
            g = mg.revcomp(g)
        return mg.bisulfite(rng, g, "CT")

    addp("rep80_pair", frag(1, 10010, 400, "+"))
    addp("rep80_pair_rc", frag(1, 10410, 400, "-"))
    addp("rep4_pair", frag(0, 5050, 300, "+"))
    addp("overlap", frag(0, 40000, 200, "+"))
    addp("full_overlap", frag(0, 41000, 150, "-"))
    addp("frag_1001", frag(0, 15000, 1001, "+"))
    addp("frag_1000", frag(0, 16000, 1000, "+"))
    for k in range(6):
        addp("near%d" % k, frag(2, 1000 + 450 * k, 420, "+"))
    while len(r1) < 500:
        ci = rng.randrange(3)
        glen = len(seqs[ci][1])
        flen = rng.randrange(150, 501)
        pos = rng.randrange(0, glen - flen + 1)
        r = rng.random()
        rate = 0.01 if r < 0.8 else (0.04 if r < 0.93 else 0.09)
        addp("p", frag(ci, pos, flen, rng.choice("+-")), rate=rate)
    mg.write_fastq(os.path.join(HERE, "pe150_1.fastq"), r1)
    mg.write_fastq(os.path.join(HERE, "pe150_2.fastq"), r2)

    tmp = tempfile.mkdtemp(prefix="walt_golden_c4_")
    idx = os.path.join(tmp, "g1.dbindex")
    mg.run([ref_makedb, "-c", os.path.join(HERE, "g1.fa"), "-o", idx], tmp)
    meta = json.load(open(os.path.join(HERE, "cases.json")))
    assert {sfx: mg.md5(idx + sfx) for sfx in ("", "_CT00", "_CT01", "_GA10", "_GA11")} == meta["index_md5"]
    cases = {"se150_ag_sam_au_m10": ("se150_ga", ["-A", "-sam", "-a", "-u", "-m", "10"]),
             "se150_ag_mr_au_m10": ("se150_ga", ["-A", "-a", "-u", "-m", "10"]),
             "pe150_sam_au_m10": ("pe150", ["-sam", "-a", "-u", "-m", "10"]),
             "pe150_mr_au_m10": ("pe150", ["-a", "-u", "-m", "10"])}
    for name, (kind, extra) in cases.items():
        wd = os.path.join(tmp, name)
        os.makedirs(wd)
        out = os.path.join(wd, "out.sam" if "-sam" in extra else "out.mr")
        cmd = [ref_walt, "-i", idx, "-o", out] + extra
        if kind.startswith("pe"):
            cmd += ["-1", os.path.join(HERE, kind + "_1.fastq"), "-2", os.path.join(HERE, kind + "_2.fastq")]
        else:
            cmd += ["-r", os.path.join(HERE, kind + ".fastq")]
        mg.run(cmd, wd)
        dst = os.path.join(HERE, "out", name)
        shutil.rmtree(dst, ignore_errors=True)
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
    print("configs[4] golden cases written:", sorted(cases))


if __name__ == "__main__":
    main()
