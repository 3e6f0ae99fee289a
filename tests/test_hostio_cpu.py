"""The driver's multi-threaded FASTQ loader (walt_amd/csrc/host/hostio.h) against its own
serial restatement and against tests/refio.py's loader, which test_oracle_golden.py pins to
the reference binary's output (batching, blank lines, dropped last character, name cutting,
srand(0)-per-batch N replacement, -C clipping).  Runs without a GPU."""
import ctypes
import os
import random

import pytest

import refio


def python_dump(path, n_per_batch, adaptor=""):
    out = []
    for names, seqs, scores in refio.load_fastq_batches(path, n_per_batch, adaptor):
        out.append("#batch %d\n" % len(names))
        for a, b, c in zip(names, seqs, scores):
            out.append("%s\t%s\t%s\n" % (a, b, c))
    return "".join(out).encode()


def cxx_dump(path, n_per_batch, adaptor, threads, serial, scratch):
    out = os.path.join(scratch, "hio_%d_%d.txt" % (threads, serial))
    used = ctypes.c_int(0)
    rc = refio.hostio_harness().hio_dump(path.encode(), n_per_batch, adaptor.encode(), threads, serial, out.encode(),
                                         ctypes.byref(used))
    assert rc == 0
    with open(out, "rb") as f:
        return f.read(), used.value


def check_all(path, n_per_batch, scratch, adaptor="", expect_serial=False):
    want = python_dump(path, n_per_batch, adaptor)
    got_serial, used = cxx_dump(path, n_per_batch, adaptor, 1, 1, scratch)
    assert used == 1
    assert got_serial == want, "serial restatement differs from refio loader"
    for threads in (1, 3, 8):
        got, used = cxx_dump(path, n_per_batch, adaptor, threads, 0, scratch)
        assert got == want, "threads=%d differs" % threads
        assert used == (1 if expect_serial else 0)


def test_sink_formatting():
    assert refio.hostio_harness().hio_format_check() == 0


@pytest.mark.parametrize("fq", ["se_ct.fastq", "se_ga.fastq", "pe_1.fastq", "pe_2.fastq", "se_clip.fastq"])
@pytest.mark.parametrize("n", [10 ** 7, 100, 7, 1])
def test_loader_golden_inputs(scratch, fq, n):
    path = os.path.join(refio.GOLDEN, fq)
    if n == 1 and os.path.getsize(path) > 300000:
        n = 3
    check_all(path, n, scratch)


def test_loader_clipping(scratch):
    meta = refio.golden_meta()
    clip = [v for v in meta["cases"].values() if v["kind"] == "se_clip"][0]
    adaptor = refio.args_to_opts(clip["args"])["C"]
    assert adaptor
    check_all(os.path.join(refio.GOLDEN, "se_clip.fastq"), 10 ** 7, scratch, adaptor=adaptor)
    check_all(os.path.join(refio.GOLDEN, "se_clip.fastq"), 50, scratch, adaptor=adaptor)


def awkward_fastq(rng, n_reads, eol=b"\n", final_newline=True, truncate=0, long_line=False):
    recs = []
    for i in range(n_reads):
        L = rng.choice([1, 2, 20, 38, 60, 100, 150, 400])
        seq = "".join(rng.choice("ACGTACGTACGTNnacgtRY.") for _ in range(L)).encode()
        qual = "".join(chr(rng.randrange(33, 75)) for _ in range(L)).encode()
        kind = rng.randrange(8)
        name = b"@r%d" % i
        if kind == 0:
            name += b" extra words here"
        elif kind == 1:
            name = b"@ lead%d x" % i       # space right after '@'
        elif kind == 2:
            name = b" sp%d tail more" % i  # line starts with a space: substr(1, npos)
        elif kind == 3:
            name = b"@"                    # empty name
        elif kind == 4:
            name += b"\tTAB kept"
        if long_line and i == n_reads // 2:
            seq = b"ACGT" * 300            # 1200 characters: fgets splits it
            qual = b"I" * 1200
        lines = [name, seq, b"+" + (name[1:] if kind == 5 else b""), qual]
        for ln in lines:
            recs.append(ln + eol)
            while rng.random() < 0.08:
                recs.append(eol if eol == b"\n" else b"\n")  # blank lines are skipped, not counted
    data = b"".join(recs)
    if truncate:
        data = data[:-truncate]
    if not final_newline and data.endswith(b"\n"):
        data = data[:-1]
    return data


@pytest.mark.parametrize("seed", range(6))
def test_loader_awkward_inputs(scratch, seed):
    rng = random.Random(seed)
    variants = [dict(), dict(final_newline=False), dict(truncate=37), dict(eol=b"\r\n"),
                dict(truncate=1), dict(final_newline=False, truncate=5)]
    data = awkward_fastq(rng, 700, **variants[seed])
    path = os.path.join(scratch, "awk_%d.fastq" % seed)
    with open(path, "wb") as f:
        f.write(data)
    for n in (10 ** 6, 64, 5):
        check_all(path, n, scratch)


def test_loader_large_multi_chunk(scratch):
    """More than one 1 MiB scan chunk per batch and several batches."""
    rng = random.Random(11)
    data = awkward_fastq(rng, 30000)
    path = os.path.join(scratch, "awk_big.fastq")
    with open(path, "wb") as f:
        f.write(data)
    assert len(data) > 5 * (1 << 20)
    for n in (10 ** 7, 9000, 29999, 30000):
        check_all(path, n, scratch)


def test_loader_long_line_takes_serial_path(scratch):
    rng = random.Random(5)
    data = awkward_fastq(rng, 400, long_line=True)
    path = os.path.join(scratch, "awk_long.fastq")
    with open(path, "wb") as f:
        f.write(data)
    check_all(path, 10 ** 6, scratch, expect_serial=True)
    # with small batches the early batches are still scanned in parallel; the result must not change
    want = python_dump(path, 50)
    got, used = cxx_dump(path, 50, "", 4, 0, scratch)
    assert got == want and used == 1


def test_loader_empty_and_tiny_files(scratch):
    for i, data in enumerate([b"", b"\n", b"@a\nACGT\n+\nIIII", b"@a\nACGT\n+\n", b"x"]):
        path = os.path.join(scratch, "tiny_%d.fastq" % i)
        with open(path, "wb") as f:
            f.write(data)
        want = python_dump(path, 10)
        for threads in (1, 4):
            got, _ = cxx_dump(path, 10, "", threads, 0, scratch)
            assert got == want


def test_clipping_of_short_reads_pinned_against_the_reference_binary(scratch, g1_index_path):
    """util.hpp:202-216 computes `s.length() - head_length + 1` in size_t: for reads of 13 bases and more this is a
    plain bound, and the loader must reproduce the reference's clipping exactly (checked through the _unmapped file
    of the REAL binary, which prints the clipped and N-replaced sequence of every too-short read); for reads of
    fewer than 13 bases the difference wraps, the reference reads far outside its string and dies -- the loader
    leaves such reads alone."""
    if not os.path.exists(refio.REF_WALT):
        pytest.skip("oracle/_ref/walt not built")
    import subprocess
    ad = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
    rng = __import__("random").Random(9)
    recs = []
    for L in (13, 14, 15, 16, 18, 20, 25, 30, 37, 38, 45, 60):
        for keep in (L, L - 3, L - 4, L - 5, L - 6, L - 9, L - 13, L - 14, 0, 1, 2):
            if keep < 0:
                continue
            body = "".join(rng.choice("ACGT") for _ in range(keep))
            tail = ad[:L - keep]
            if len(tail) > 5 and rng.random() < 0.5:  # one error inside the adaptor part
                k = rng.randrange(len(tail))
                tail = tail[:k] + rng.choice([c for c in "ACGT" if c != tail[k]]) + tail[k + 1:]
            recs.append(("r%d_%d_%d" % (len(recs), L, keep), (body + tail)[:L]))
    fq = os.path.join(scratch, "clip_short.fastq")
    with open(fq, "w") as f:
        for nm, sq in recs:
            f.write("@%s\n%s\n+\n%s\n" % (nm, sq, "I" * len(sq)))
    wd = os.path.join(scratch, "clip_short_ref")
    os.makedirs(wd, exist_ok=True)
    out = os.path.join(wd, "o.mr")
    subprocess.run([refio.REF_WALT, "-i", g1_index_path, "-r", fq, "-o", out, "-C", ad, "-u", "-a", "-m", "0"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref_seq = {}
    for side in ("_unmapped", "_ambiguous", ""):
        for ln in open(out + side):
            f = ln.rstrip("\n").split("\t")
            if side == "_unmapped":
                ref_seq[f[0]] = f[1]
            else:
                ref_seq[f[3]] = f[6] if f[5] == "+" else refio.revcomp(f[6])
    got = cxx_dump(fq, 10 ** 6, ad, 3, 0, scratch)[0].decode()
    mine = {ln.split("\t")[0]: ln.split("\t")[1] for ln in got.splitlines() if not ln.startswith("#")}
    assert len(ref_seq) == len(recs) and sorted(mine) == sorted(ref_seq)
    clipped = 0
    for (nm, sq) in recs:
        assert mine[nm] == ref_seq[nm], (nm, sq, mine[nm], ref_seq[nm])
        clipped += mine[nm] != sq
    assert clipped > 40
    # below 13 bases: the reference binary does not survive the wrap; the loader does, and clips nothing
    fq2 = os.path.join(scratch, "clip_tiny.fastq")
    with open(fq2, "w") as f:
        f.write("@t0\nACGTAGATCGGA\n+\nIIIIIIIIIIII\n@t1\nAGATCGG\n+\nIIIIIII\n")
    r = subprocess.run([refio.REF_WALT, "-i", g1_index_path, "-r", fq2, "-o", os.path.join(wd, "t.mr"), "-C", ad, "-u"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert r.returncode != 0, "the reference was expected to die on the unsigned wrap (util.hpp:203)"
    got = cxx_dump(fq2, 10 ** 6, ad, 2, 0, scratch)[0].decode()
    assert [ln.split("\t")[1] for ln in got.splitlines() if not ln.startswith("#")] == ["ACGTAGATCGGA", "AGATCGG"]
