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
