#!/usr/bin/env python3
"""Golden fixtures for the reference's other seed patterns (5 and 7).

Run in the build container only: needs oracle/_ref/{makedb,walt}_sp{5,7}, the reference rebuilt with
-D SEEDPATTERN5 / -D SEEDPATTERN7 (oracle/Makefile.ref; FAQ.md:5-13 of the reference).  Commits DATA only:
  sp_*.fastq                 the read sets of make_golden.py restricted to reads of at most 148 bases -- with
                             patterns 5 / 7 the reference indexes its seed tables out of bounds for longer
                             reads (mapping.cpp:238 caps the repeats at 50, the tables hold 28 / 20)
  seedpattern{5,7}.json      data dump of the header's tables for these patterns
  cases_sp{5,7}.json         every reference invocation + md5 of the five index files
  out_sp{5,7}/<case>/*.gz    the reference's outputs
"""
import gzip
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import md5, run  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(HERE))
REF_BIN = os.path.join(ROOT, "oracle", "_ref")
MAXLEN = 148


def fastq_records(path):
    with open(path) as f:
        lines = f.read().split("\n")
    return [lines[i:i + 4] for i in range(0, len(lines) - 3, 4)]


def filter_se(src, dst):
    recs = [r for r in fastq_records(os.path.join(HERE, src)) if len(r[1]) <= MAXLEN]
    with open(os.path.join(HERE, dst), "w") as f:
        for r in recs:
            f.write("\n".join(r) + "\n")
    return len(recs)


def filter_pe(src1, src2, dst1, dst2):
    a, b = fastq_records(os.path.join(HERE, src1)), fastq_records(os.path.join(HERE, src2))
    keep = [i for i in range(len(a)) if len(a[i][1]) <= MAXLEN and len(b[i][1]) <= MAXLEN]
    for src, dst in ((a, dst1), (b, dst2)):
        with open(os.path.join(HERE, dst), "w") as f:
            for i in keep:
                f.write("\n".join(src[i]) + "\n")
    return len(keep)


def dump_tables(pat, hdr="/root/reference/src/walt/seedpattern.hpp"):
    src = open(hdr).read()
    blk = src[src.index("#ifdef SEEDPATTERN%d" % pat):]
    blk = blk[:blk.index("#endif")]

    def body(name):
        m = re.search(name + r"\[[^\]]*\](?:\[[^\]]*\])?\s*=\s*\{(.*?)\};", blk, re.S)
        return re.sub(r"/\*.*?\*/", "", m.group(1))

    dims = re.search(r"F2NOCAREDPOSITION\[(\d+)\]\[(\d+)\]", blk)
    width = int(dims.group(2))
    care = [int(x) for x in re.findall(r"\d+", body("F2CAREDPOSITION"))]
    nocare = []
    for r in re.findall(r"\{([^{}]*)\}", body("F2NOCAREDPOSITION")):
        v = [int(x) for x in re.findall(r"\d+", r)]
        nocare.append(v + [0] * (width - len(v)))  # C++ zero-fills the rest of each row
    with open(os.path.join(HERE, "seedpattern%d.json" % pat), "w") as f:
        json.dump({"F2CAREDPOSITION": care, "F2NOCAREDPOSITION": nocare}, f)


def make_short_reads(path):
    """Reads at the lower end of the length range: MINIMALREADLEN is 32 / 23 for patterns 5 / 7, and reads of
    32-33 (pattern 5) or 25-26 (pattern 7) bases have seeds of 10 / 8 care characters, fewer than the 12 that
    getHashValue reads (util.hpp:175-182).  23- and 24-base reads are left out: with pattern 7 the reference
    hashes beyond their end and exits in getBits (util.hpp:117-119) as soon as it reaches seed shift 5."""
    import random
    rng = random.Random(5577)
    seqs, name, cur = [], None, []
    for line in open(os.path.join(HERE, "g1.fa")):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if name is not None:
                seqs.append("".join(cur).upper())
            name, cur = line, []
        else:
            cur.append(line)
    seqs.append("".join(cur).upper())
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    with open(path, "w") as f:
        k = 0
        for L in list(range(25, 41)) * 12:
            g = seqs[rng.randrange(3)]
            p = rng.randrange(0, len(g) - L + 1)
            s = g[p:p + L]
            if rng.random() < 0.5:
                s = "".join(comp[c] for c in reversed(s))
            s = "".join("T" if (c == "C" and rng.random() < 0.95) else c for c in s)
            if rng.random() < 0.4:
                i = rng.randrange(L)
                s = s[:i] + rng.choice([c for c in "ACGT" if c != s[i]]) + s[i + 1:]
            f.write("@short_%d_len%d\n%s\n+\n%s\n" % (k, L, s, "I" * L))
            k += 1


CASES = {
    "se_short_sam_au": ("sp_short", ["-sam", "-a", "-u"]),
    "se_short_mr_au_m2": ("sp_short", ["-a", "-u", "-m", "2"]),
    "se_mr": ("sp_se_ct", []),
    "se_sam_au": ("sp_se_ct", ["-sam", "-a", "-u"]),
    "se_sam_au_m10": ("sp_se_ct", ["-sam", "-a", "-u", "-m", "10"]),
    "se_sam_au_b2": ("sp_se_ct", ["-sam", "-a", "-u", "-b", "2"]),
    "se_ag_sam_au": ("sp_se_ga", ["-A", "-sam", "-a", "-u"]),
    "pe_sam_au": ("sp_pe", ["-sam", "-a", "-u"]),
    "pe_mr_au": ("sp_pe", ["-a", "-u"]),
    "pe_sam_au_k3": ("sp_pe", ["-sam", "-a", "-u", "-k", "3"]),
    "pe_sam_au_m10_b20": ("sp_pe", ["-sam", "-a", "-u", "-m", "10", "-b", "20"]),
}


def main():
    n1 = filter_se("se_ct.fastq", "sp_se_ct.fastq")
    n2 = filter_se("se_ga.fastq", "sp_se_ga.fastq")
    n3 = filter_pe("pe_1.fastq", "pe_2.fastq", "sp_pe_1.fastq", "sp_pe_2.fastq")
    make_short_reads(os.path.join(HERE, "sp_short.fastq"))
    print("read sets: %d + %d single-end, %d pairs" % (n1, n2, n3))
    for pat in (5, 7):
        sfx = "_sp%d" % pat
        if not os.path.exists(os.path.join(REF_BIN, "walt" + sfx)):
            sys.exit("build oracle/_ref first: make -f oracle/Makefile.ref")
        dump_tables(pat)
        tmp = tempfile.mkdtemp(prefix="walt_golden%s_" % sfx)
        idx = os.path.join(tmp, "g1.dbindex")
        run([os.path.join(REF_BIN, "makedb" + sfx), "-c", os.path.join(HERE, "g1.fa"), "-o", idx], tmp)
        meta = {"index_md5": {s: md5(idx + s) for s in ("", "_CT00", "_CT01", "_GA10", "_GA11")}, "cases": {}}
        outroot = os.path.join(HERE, "out" + sfx)
        shutil.rmtree(outroot, ignore_errors=True)
        for name, (kind, extra) in CASES.items():
            wd = os.path.join(tmp, name)
            os.makedirs(wd)
            out = os.path.join(wd, "out.sam" if "-sam" in extra else "out.mr")
            cmd = [os.path.join(REF_BIN, "walt" + sfx), "-i", idx, "-o", out] + extra
            if kind == "sp_pe":
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
        with open(os.path.join(HERE, "cases%s.json" % sfx), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        shutil.rmtree(tmp)
        print("pattern %d: %d cases written" % (pat, len(CASES)))


if __name__ == "__main__":
    main()
