"""GPU makedb (walt_index_build_device): same index as the reference makedb /
the host builder, up to the order of entries whose 60 care characters are all
equal; mapping through a device-built index equals the oracle on the exported
arrays; written .dbindex files are readable by the real reference binary."""
import os
import random
import subprocess

import numpy as np
import pytest

import refio
from test_harness_cpu import assert_best_equal, make_random_case, sample_reads

pytestmark = pytest.mark.gpu


def upload_genome(torch, db_or_seqs):
    s = "".join(x[1].upper() for x in db_or_seqs)
    t = torch.frombuffer(bytearray(s.encode()), dtype=torch.uint8).cuda()
    return t, [len(x[1]) for x in db_or_seqs], [x[0] for x in db_or_seqs]


def read_fasta(path):
    seqs, name, cur = [], None, []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if name is not None:
                seqs.append((name, "".join(cur)))
            name, cur = line[1:].split()[0].split("\t")[0], []
        else:
            cur.append(line)
    seqs.append((name, "".join(cur)))
    return seqs


def canonical_order(db, s):
    """index[] of a host-built strand re-ordered inside every run of equal
    (bucket, 48 marked care chars) by ascending position."""
    g = db.genome[s]
    ix = db.index[s].astype(np.int64)
    start = db.start_index.astype(np.int64)
    chr_ = np.searchsorted(start, ix, side="right") - 1
    room = start[chr_ + 1] - ix
    code = np.zeros(256, dtype=np.int64)
    code[ord("A")], code[ord("C")], code[ord("G")], code[ord("T")] = 1, 2, 2, 3
    gpad = np.concatenate([g, np.zeros(200, dtype=np.uint8)])
    cols = []
    for q in range(12, 60):
        cp = 1 + 3 * q
        cols.append(np.where(cp < room, code[gpad[ix + cp]], 0))
    bucket = np.repeat(np.arange(1 << 24, dtype=np.int64), np.diff(db.counter[s].astype(np.int64)))
    keys = [ix] + cols[::-1] + [bucket]  # lexsort: last key is primary
    order = np.lexsort(keys)
    return db.index[s][order]


@pytest.mark.parametrize("which", ["g1", "random"])
def test_gpu_builder_matches_host_builder(scratch, g1_db, which):
    import torch
    import walt_amd
    if which == "g1":
        seqs = read_fasta(os.path.join(refio.GOLDEN, "g1.fa"))
        db = g1_db
    else:
        seqs, db = make_random_case(41, 40, scratch)
    d_g, lens, names = upload_genome(torch, seqs)
    idx = walt_amd.Index.build_device(d_g.data_ptr(), lens, names, device=0)
    assert idx.chrom_names == db.names and idx.chrom_lengths == db.lengths.tolist()
    for s in range(4):
        g, cnt, ix = idx.export_strand(s)
        assert np.array_equal(g, db.genome[s]), "strand %d genome" % s
        assert np.array_equal(cnt, db.counter[s]), "strand %d counter" % s
        assert ix.size == db.index[s].size
        assert np.array_equal(ix, canonical_order(db, s)), "strand %d index order" % s
        if which == "random" and np.array_equal(db.index[s], canonical_order(db, s)):
            assert np.array_equal(ix, db.index[s])  # tie-free: byte-identical to the reference makedb
    # mapping through the device-built index == oracle on the exported arrays
    rng = random.Random(7)
    reads = sample_reads(rng, [(n, s.upper()) for n, s in seqs if len(s) >= 40], 1500, "CT")
    out_path = os.path.join(scratch, "devbuilt_%s.dbindex" % which)
    idx.write(out_path)
    db2 = refio.DbIndex(out_path)
    want, _ = refio.oracle_se(db2, reads)
    got, _ = idx.map_se_batch(*walt_amd.pack_reads(reads))
    assert_best_equal(got, want, "device-built index")
    # the real reference binary accepts the written files and agrees
    if os.path.exists(refio.REF_WALT):
        fq = os.path.join(scratch, "devbuilt_%s.fastq" % which)
        with open(fq, "w") as f:
            for i, r in enumerate(reads):
                f.write("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)))
        sam = os.path.join(scratch, "devbuilt_%s.sam" % which)
        subprocess.run([refio.REF_WALT, "-i", out_path, "-r", fq, "-o", sam, "-sam", "-a", "-u"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        mine = refio.sam_header(db2)
        for i, (rec, r) in enumerate(zip(got, reads)):
            mine += refio.se_sam_line(db2, rec, "r%d" % i, r, "I" * len(r), True, True)
        assert open(sam).read() == mine
    idx.close()
