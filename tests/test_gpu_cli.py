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


@pytest.mark.parametrize("env", [{"se_pipe": "0"}, {"se_pipe": "0", "se_heavy_chunk": "64"},
                                 {"se_heavy_chunk": "64"}, {"se_lit_side": "0"}, {"se_lit_side": "1"}, {"se_lit_side": "2"}, {"se_lit_side": "2", "se_pipe": "0"}, {"se_lit_staged": "1"}, {"se_carry": "0", "se_heavy_chunk": "64"}])
@pytest.mark.parametrize("case", ["se_sam_au", "se150_ag_sam_au_m10"])
def test_cli_single_end_schedules_give_the_same_files(cli_index, scratch, case, env):
    """The staged heavy pass has two schedules -- two halves of the heavy list on two streams, each with its own state
    slot and the list cut evenly on the device (default), or one stream in chunks (option se_pipe = 0) -- the literal
    pass runs beside its end, after it (se_lit_side = 0) or as staged rounds (se_lit_staged = 1), and the staged rounds go on from pass 1's state or start over
    (se_carry = 0).  Each combination through the command line (-X name=value), with chunks of 64 reads so that several
    chunks, both slots and the later chunks' wait for the literal snapshot are all exercised; the files must be the
    reference's."""
    if case not in META["cases"]:
        pytest.skip("no golden case " + case)
    info = META["cases"][case]
    wd = os.path.join(scratch, "cli_sched_%s_%s" % (case, "_".join(sorted(env)) + "".join(env.values())))
    os.makedirs(wd, exist_ok=True)
    out = os.path.join(wd, "out.sam" if "-sam" in info["args"] else "out.mr")
    cmd = [WALT_BIN, "-i", cli_index, "-o", out] + list(info["args"]) + ["-r", os.path.join(refio.GOLDEN, info["kind"] + ".fastq")]
    for k, v in sorted(env.items()):
        cmd += ["-X", "%s=%s" % (k, v)]
    subprocess.run(cmd, check=True, cwd=wd, stderr=subprocess.DEVNULL)
    assert sorted(os.listdir(wd)) == sorted(info["files"])
    for fn in info["files"]:
        with open(os.path.join(wd, fn)) as f:
            assert f.read() == refio.golden_file(case, fn), "%s/%s differs under %s" % (case, fn, env)


@pytest.mark.parametrize("case", ["se_sam_au_N100", "se_ag_mr_au", "se_clip_mr_au_N100", "pe_sam_au_N250", "pe_mr_au", "pe150_sam_au_m10"])
def test_cli_several_gpus_share_each_batch(cli_index, scratch, case):
    """-g 0,0: two index replicas (here on the one GPU of the box), every -N batch cut into two contiguous shares
    after the loader's N replacement, mapped side by side, records back in read order, statistics added on the
    host: the files must still be the reference's, byte for byte."""
    info = META["cases"][case]
    wd = os.path.join(scratch, "cli2_" + case)
    os.makedirs(wd, exist_ok=True)
    out = os.path.join(wd, "out.sam" if "-sam" in info["args"] else "out.mr")
    cmd = [WALT_BIN, "-i", cli_index, "-o", out, "-g", "0,0", "-t", "5"] + list(info["args"])
    kind = info["kind"]
    if kind.startswith("pe"):
        cmd += ["-1", os.path.join(refio.GOLDEN, kind + "_1.fastq"), "-2", os.path.join(refio.GOLDEN, kind + "_2.fastq")]
    else:
        cmd += ["-r", os.path.join(refio.GOLDEN, kind + ".fastq")]
    subprocess.run(cmd, check=True, cwd=wd, stderr=subprocess.DEVNULL)
    assert sorted(os.listdir(wd)) == sorted(info["files"])
    for fn in info["files"]:
        with open(os.path.join(wd, fn)) as f:
            assert f.read() == refio.golden_file(case, fn), "%s/%s differs with -g 0,0" % (case, fn)


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


def _run_cli(wd, args):
    os.makedirs(wd, exist_ok=True)
    for fn in os.listdir(wd):
        os.remove(os.path.join(wd, fn))
    subprocess.run([WALT_BIN] + args, check=True, cwd=wd, stderr=subprocess.DEVNULL)
    return {fn: open(os.path.join(wd, fn)).read() for fn in sorted(os.listdir(wd))}


def _swap_mapstats_mates(text):
    lines = text.split("\n")
    i1, i2 = lines.index("mate1:"), lines.index("mate2:")
    i3 = lines.index("frag_len_distribution:")
    return "\n".join(lines[:i1] + ["mate1:"] + lines[i2 + 1:i3] + ["mate2:"] + lines[i1 + 1:i2] + lines[i3:])


@pytest.mark.parametrize("kind,sam_case,mr_case,extra", [("pe", "pe_sam_au", "pe_mr_au", []),
                                                         ("pe150", "pe150_sam_au_m10", "pe150_mr_au_m10", ["-m", "10"])])
def test_cli_pbat_is_the_mate_swapped_run_put_back_in_user_order(cli_index, scratch, kind, sam_case, mr_case, extra):
    """-P has no implementation in the reference snapshot (parity unpinned).  It is defined by
    equivalence: `-P -1 X -2 Y` maps like `-1 Y -2 X`, then restores the user's order: X's record first,
    0x40 on X / 0x80 on Y, QNAME from X, _1 side files and the mate1 mapstats block for X.  So with
    X = pe_2.fastq, Y = pe_1.fastq the expected files are a rewrite of the golden pe_1/pe_2 outputs.
    The second parameter set is BASELINE configs[4]: 2 x 150 bp, -m 10, PBAT."""
    p1, p2 = os.path.join(refio.GOLDEN, kind + "_1.fastq"), os.path.join(refio.GOLDEN, kind + "_2.fastq")

    def qname(n):  # golden runs print pe_1's names ("…/1"); the PBAT run prints its own -1 file's ("…/2")
        assert n.endswith("/1")
        return n[:-1] + "2"

    # SAM
    got = _run_cli(os.path.join(scratch, "pbat_sam_" + kind), ["-i", cli_index, "-o", "out.sam", "-P", "-1", p2, "-2", p1,
                                                              "-sam", "-a", "-u"] + extra)
    want_lines = refio.golden_file(sam_case, "out.sam").splitlines()
    head = [ln for ln in want_lines if ln.startswith("@")]
    body = [ln for ln in want_lines if not ln.startswith("@")]
    assert len(body) % 2 == 0
    exp = list(head)
    for a, b in zip(body[0::2], body[1::2]):
        fa, fb = a.split("\t"), b.split("\t")
        assert int(fa[1]) & 0x40 and int(fb[1]) & 0x80
        fa[1], fb[1] = str(int(fa[1]) ^ 0xC0), str(int(fb[1]) ^ 0xC0)
        fa[0], fb[0] = qname(fa[0]), qname(fb[0])
        exp += ["\t".join(fb), "\t".join(fa)]
    assert got["out.sam"].splitlines() == exp
    assert got["out.sam.mapstats"] == _swap_mapstats_mates(refio.golden_file(sam_case, "out.sam.mapstats"))

    # MR with side files
    got = _run_cli(os.path.join(scratch, "pbat_mr_" + kind), ["-i", cli_index, "-o", "out.mr", "-P", "-1", p2, "-2", p1, "-a", "-u"] + extra)

    def rename(line, col):
        f = line.split("\t")
        f[col] = ("FRAG:" + qname(f[col][5:])) if f[col].startswith("FRAG:") else qname(f[col])
        return "\t".join(f)

    want_main = [rename(ln, 3) for ln in refio.golden_file(mr_case, "out.mr").splitlines()]
    got_main = got["out.mr"].splitlines()
    frag = lambda ls: [ln for ln in ls if "\tFRAG:" in ln]
    single = lambda ls: sorted(ln for ln in ls if "\tFRAG:" not in ln)
    assert frag(got_main) == frag(want_main) and single(got_main) == single(want_main)
    assert len(frag(got_main)) > 100
    for mine, theirs in (("1", "2"), ("2", "1")):
        for side, col in (("ambiguous", 3), ("unmapped", 0)):
            want = [rename(ln, col) for ln in refio.golden_file(mr_case, "out.mr_%s_%s" % (theirs, side)).splitlines()]
            assert got["out.mr_%s_%s" % (mine, side)].splitlines() == want, (mine, side)
    assert got["out.mr.mapstats"] == _swap_mapstats_mates(refio.golden_file(mr_case, "out.mr.mapstats"))

    # single-end: -P is -A
    ga = os.path.join(refio.GOLDEN, "se150_ga.fastq" if kind == "pe150" else "se_ga.fastq")
    a = _run_cli(os.path.join(scratch, "pbat_se_a"), ["-i", cli_index, "-o", "o.sam", "-A", "-r", ga, "-sam", "-a", "-u"] + extra)
    b = _run_cli(os.path.join(scratch, "pbat_se_p"), ["-i", cli_index, "-o", "o.sam", "-P", "-r", ga, "-sam", "-a", "-u"] + extra)
    assert a == b and len(a["o.sam"]) > 1000
    if kind == "pe150":  # and that is the reference's own -A output for configs[4]'s single-end half
        assert a["o.sam"] == refio.golden_file("se150_ag_sam_au_m10", "out.sam")


def test_cli_makedb_on_gpu_writes_files_the_reference_binary_maps_from(scratch):
    """`makedb -g 0`: FASTA -> GPU builder -> .dbindex files.  The golden genome has entries whose 60 compared
    characters tie, so the index may differ from the reference's in the order of those entries only: head,
    genome and counter sections must be identical to the host builder's files, and both binaries must produce
    the same SAM from the GPU-built files."""
    fa = os.path.join(refio.GOLDEN, "g1.fa")
    host, dev = os.path.join(scratch, "mk_host.dbindex"), os.path.join(scratch, "mk_dev.dbindex")
    env = dict(os.environ, WALT_MAKEDB_SEED="1")
    subprocess.run([MAKEDB_BIN, "-c", fa, "-o", host, "-t", "4"], check=True, env=env, stderr=subprocess.DEVNULL)
    subprocess.run([MAKEDB_BIN, "-c", fa, "-o", dev, "-g", "0"], check=True, env=env, stderr=subprocess.DEVNULL)
    assert open(host, "rb").read() == open(dev, "rb").read()
    a, b = refio.DbIndex(host), refio.DbIndex(dev)
    for s in range(4):
        assert (a.genome[s] == b.genome[s]).all() and (a.counter[s] == b.counter[s]).all()
        assert sorted(a.index[s].tolist()) == sorted(b.index[s].tolist())
    if not os.path.exists(refio.REF_WALT):
        pytest.skip("oracle/_ref/walt not built")
    outs = []
    for who, binary in (("gpu", WALT_BIN), ("ref", refio.REF_WALT)):
        o = os.path.join(scratch, "mk_%s.sam" % who)
        if os.path.exists(o):
            os.remove(o)
        subprocess.run([binary, "-i", dev, "-r", os.path.join(refio.GOLDEN, "se_ct.fastq"), "-o", o, "-sam", "-a", "-u"],
                       check=True, stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
        outs.append(open(o, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 1000
