"""End-to-end drop-in check: the product's `walt` binary (walt_amd/bin/walt: WALT's CLI,
FASTQ loader, adaptor clipping and SAM/MR/mapstats writers on top of the C ABI) is run
with the reference's own arguments on the golden inputs; every output file must be
byte-identical to what the REAL reference binary wrote (tests/golden/out/)."""
import os
import subprocess

import pytest

import refio
from test_oracle_golden import META

pytestmark = pytest.mark.gpu

WALT_BIN = os.path.join(refio.ROOT, "walt_amd", "bin", "walt")
MAKEDB_BIN = os.path.join(refio.ROOT, "walt_amd", "bin", "makedb")


@pytest.fixture(scope="module")
def cli_index(scratch):
    out = os.path.join(scratch, "cli_g1.dbindex")
    env = dict(os.environ, WALT_MAKEDB_SEED="1")
    subprocess.run([MAKEDB_BIN, "-c", os.path.join(refio.GOLDEN, "g1.fa"), "-o", out, "-t", "4"], check=True, env=env,
                   stderr=subprocess.DEVNULL)
    return out


@pytest.mark.parametrize("case", sorted(META["cases"]))
def test_cli_output_files_identical_to_reference(cli_index, scratch, case):
    info = META["cases"][case]
    wd = os.path.join(scratch, "cli_" + case)
    os.makedirs(wd, exist_ok=True)
    sam = "-sam" in info["args"]
    out = os.path.join(wd, "out.sam" if sam else "out.mr")
    cmd = [WALT_BIN, "-i", cli_index, "-o", out] + list(info["args"])
    kind = info["kind"]
    if kind.startswith("pe"):
        cmd += ["-1", os.path.join(refio.GOLDEN, kind + "_1.fastq"), "-2", os.path.join(refio.GOLDEN, kind + "_2.fastq")]
    else:
        cmd += ["-r", os.path.join(refio.GOLDEN, kind + ".fastq")]
    subprocess.run(cmd, check=True, cwd=wd, stderr=subprocess.DEVNULL)
    produced = sorted(os.listdir(wd))
    assert produced == sorted(info["files"]), (produced, info["files"])
    for fn in info["files"]:
        with open(os.path.join(wd, fn)) as f:
            got = f.read()
        want = refio.golden_file(case, fn)
        if got != want:
            for i, (a, b) in enumerate(zip(got.splitlines(), want.splitlines())):
                assert a == b, "%s/%s line %d:\n got: %s\nwant: %s" % (case, fn, i + 1, a, b)
            assert len(got.splitlines()) == len(want.splitlines()), "%s/%s line count" % (case, fn)
            assert False, "%s/%s differs" % (case, fn)


def test_cli_rejects_bad_arguments(cli_index, scratch):
    r = subprocess.run([WALT_BIN, "-i", cli_index, "-r", os.path.join(refio.GOLDEN, "se_ct.fastq"), "-o",
                        os.path.join(scratch, "x.sam"), "-k", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "paired-end candidates must be in [2, 300]" in r.stderr  # walt.cpp:245-246
    r = subprocess.run([WALT_BIN, "-i", "/nonexistent.dbindex", "-r", "a.fastq", "-o", "o"], capture_output=True, text=True)
    assert r.returncode != 0


def _awkward_copy(src, dst, seed):
    """se/pe golden reads with Ns, lower case, blank lines, decorated names and no final newline."""
    import random
    rng = random.Random(seed)
    with open(src, "rb") as f:
        lines = f.read().split(b"\n")
    out = []
    for i in range(0, len(lines) - 3, 4):
        name, seq, plus, qual = lines[i:i + 4]
        seq = bytearray(seq)
        for _ in range(rng.choice([0, 0, 1, 2, 5])):
            seq[rng.randrange(len(seq))] = rng.choice(b"Nnacgt.")
        k = rng.randrange(4)
        if k == 0:
            name += b" 1:N:0:ACGT"
        elif k == 1:
            name = b"@ " + name[1:]
        for ln in (name, bytes(seq), plus, qual):
            out.append(ln + b"\n")
            if rng.random() < 0.05:
                out.append(b"\n")
    data = b"".join(out)[:-1]
    with open(dst, "wb") as f:
        f.write(data)


@pytest.mark.parametrize("mode,extra", [("se", ["-N", "10000000"]), ("se", ["-N", "37", "-sam"]), ("se", ["-N", "1000", "-A"]),
                                         ("pe", ["-N", "10000000", "-sam"]), ("pe", ["-N", "53"])])
@pytest.mark.parametrize("threads", ["1", "6"])
def test_cli_awkward_fastq_matches_reference_binary(cli_index, scratch, mode, extra, threads):
    """Both binaries on the same awkward FASTQ: exercises the threaded loader's batch boundaries
    and the srand(0)-per-batch N replacement end to end."""
    if not os.path.exists(refio.REF_WALT):
        pytest.skip("oracle/_ref/walt not built")
    tag = "%s_%s_%s" % (mode, "_".join(a.strip("-") for a in extra), threads)
    wd = os.path.join(scratch, "awk_" + tag)
    os.makedirs(wd, exist_ok=True)
    if mode == "se":
        fq = os.path.join(wd, "reads.fastq")
        _awkward_copy(os.path.join(refio.GOLDEN, "se_ga.fastq" if "-A" in extra else "se_ct.fastq"), fq, 3)
        inputs = ["-r", fq]
    else:
        f1, f2 = os.path.join(wd, "r_1.fastq"), os.path.join(wd, "r_2.fastq")
        _awkward_copy(os.path.join(refio.GOLDEN, "pe_1.fastq"), f1, 4)
        _awkward_copy(os.path.join(refio.GOLDEN, "pe_2.fastq"), f2, 5)
        inputs = ["-1", f1, "-2", f2]
    outs = {}
    for who, binary in (("gpu", WALT_BIN), ("ref", refio.REF_WALT)):
        od = os.path.join(wd, who)
        os.makedirs(od, exist_ok=True)
        for fn in os.listdir(od):
            os.remove(os.path.join(od, fn))
        subprocess.run([binary, "-i", cli_index, "-o", os.path.join(od, "out"), "-a", "-u", "-t", threads] + inputs + extra,
                       check=True, stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
        outs[who] = {fn: open(os.path.join(od, fn), "rb").read() for fn in sorted(os.listdir(od))}
    assert sorted(outs["gpu"]) == sorted(outs["ref"])
    for fn in outs["ref"]:
        assert outs["gpu"][fn] == outs["ref"][fn], "%s: %s differs" % (tag, fn)
    assert len(outs["ref"]["out"]) > 1000
