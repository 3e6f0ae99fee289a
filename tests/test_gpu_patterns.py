"""Seed patterns 5 and 7 on the GPU (libwalt_amd_sp5.so / _sp7.so, bin/walt_sp5 / walt_sp7):
  * single-end and paired-end mapping through the C ABI reproduce the files the real reference binaries
    (rebuilt with -D SEEDPATTERN5 / 7) wrote for the golden cases, and equal the oracle record by record;
  * the GPU index builder equals the host builder (md5-identical to the reference makedb, test_patterns_cpu)
    up to the order of entries whose care characters are all equal;
  * the command-line binaries write byte-identical files;
  * a read longer than the pattern's tables cover is refused (the reference indexes out of bounds there)."""
import os
import subprocess

import numpy as np
import pytest

import refio
from test_harness_cpu import assert_best_equal
from test_oracle_golden import check_against_golden, run_pe_case, run_se_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=(5, 7))
def pat(request, scratch):
    import walt_amd
    p = request.param
    refio.set_pattern(p)
    walt_amd.set_pattern(p)
    try:
        path = os.path.join(scratch, "g1_gpu_sp%d.dbindex" % p)
        walt_amd.makedb(os.path.join(refio.GOLDEN, "g1.fa"), path, threads=4)
        db = refio.DbIndex(path)
        idx = walt_amd.Index.open(path, device=0)
        yield p, refio.golden_meta(), db, idx, path
        idx.close()
    finally:
        refio.set_pattern(3)
        walt_amd.set_pattern(3)


@pytest.mark.parametrize("case", ["se_mr", "se_sam_au", "se_sam_au_m10", "se_sam_au_b2", "se_ag_sam_au",
                                  "se_short_sam_au", "se_short_mr_au_m2"])
def test_gpu_se_reproduces_reference_files(pat, case):
    import walt_amd
    _, meta, db, idx, _ = pat

    def mapper(seqs, ag, m, b):
        got, stats = idx.map_se_batch(*walt_amd.pack_reads(seqs), ag_wildcard=ag, max_mismatches=m, b=b)
        want, work = refio.oracle_se(db, seqs, ag=ag, max_mm=m, b=b)
        assert_best_equal(got, want, case)
        assert int(stats["too_short"]) == int(work["too_short"])
        return got, int(stats["too_short"])
    check_against_golden(case, run_se_case(db, case, mapper, meta), meta)


@pytest.mark.parametrize("case", ["pe_sam_au", "pe_mr_au", "pe_sam_au_k3", "pe_sam_au_m10_b20"])
def test_gpu_pe_reproduces_reference_files(pat, case):
    import walt_amd
    _, meta, db, idx, _ = pat

    def mapper(s1, s2, m, b, k, L):
        res, stats, ranked = idx.map_pe_batch(*walt_amd.pack_reads(s1), *walt_amd.pack_reads(s2), max_mismatches=m,
                                              b=b, top_k=k, frag_range=L, want_ranked=True)
        wantp, wranked, _ = refio.oracle_pe(db, s1, s2, max_mm=m, b=b, top_k=k, frag_range=L)
        for f in ("best_times", "frag_len", "pair_mm", "best_i", "best_j"):
            assert np.array_equal(res[f], wantp[f]), f
        assert np.array_equal(ranked[1], wranked[1]) and np.array_equal(ranked[3], wranked[3])
        return res, ranked, (int(stats[0]["too_short"]), int(stats[1]["too_short"]))
    check_against_golden(case, run_pe_case(db, case, mapper, meta), meta)


def test_gpu_builder_matches_host_builder(pat):
    import torch
    import walt_amd
    from test_gpu_builder import read_fasta, upload_genome
    p, _, db, _, _ = pat
    seqs = read_fasta(os.path.join(refio.GOLDEN, "g1.fa"))
    d_g, lens, names = upload_genome(torch, seqs)
    idx = walt_amd.Index.build_device(d_g.data_ptr(), lens, names, device=0)
    ncare = {5: 56, 7: 80}[p]
    care = [(i // 2) * 5 + (i % 2) * 2 for i in range(ncare)] if p == 5 else \
           [(i // 4) * 7 + (0, 1, 2, 4)[i % 4] for i in range(ncare)]
    start = db.start_index.astype(np.int64)
    code = np.zeros(256, dtype=np.int64)
    code[ord("A")], code[ord("C")], code[ord("G")], code[ord("T")] = 1, 2, 2, 3
    for s in range(4):
        g, cnt, ix = idx.export_strand(s)
        assert np.array_equal(g, db.genome[s]) and np.array_equal(cnt, db.counter[s])
        # same entries in every run of equal (bucket, marked care characters); inside a run the host builder
        # keeps what std::sort leaves (= the reference's order), the GPU builder ascending positions
        hix = db.index[s].astype(np.int64)
        chr_ = np.searchsorted(start, hix, side="right") - 1
        room = start[chr_ + 1] - hix
        gpad = np.concatenate([db.genome[s], np.zeros(300, dtype=np.uint8)])
        cols = [np.where(care[q] < room, code[gpad[hix + care[q]]], 0) for q in range(12, ncare)]
        bucket = np.repeat(np.arange(1 << 24, dtype=np.int64), np.diff(db.counter[s].astype(np.int64)))
        order = np.lexsort([hix] + cols[::-1] + [bucket])
        assert np.array_equal(ix, db.index[s][order]), "strand %d" % s
    idx.close()


def test_cli_output_files_identical_to_reference(pat, scratch):
    p, meta, _, _, _ = pat
    walt_bin = os.path.join(refio.ROOT, "walt_amd", "bin", "walt_sp%d" % p)
    makedb_bin = os.path.join(refio.ROOT, "walt_amd", "bin", "makedb_sp%d" % p)
    index = os.path.join(scratch, "cli_g1_sp%d.dbindex" % p)
    subprocess.run([makedb_bin, "-c", os.path.join(refio.GOLDEN, "g1.fa"), "-o", index, "-t", "4"], check=True,
                   env=dict(os.environ, WALT_MAKEDB_SEED="1"), stderr=subprocess.DEVNULL)
    for case, info in sorted(meta["cases"].items()):
        wd = os.path.join(scratch, "cli_sp%d_%s" % (p, case))
        os.makedirs(wd, exist_ok=True)
        out = os.path.join(wd, "out.sam" if "-sam" in info["args"] else "out.mr")
        cmd = [walt_bin, "-i", index, "-o", out] + list(info["args"])
        kind = info["kind"]
        if kind == "sp_pe":
            cmd += ["-1", os.path.join(refio.GOLDEN, kind + "_1.fastq"), "-2", os.path.join(refio.GOLDEN, kind + "_2.fastq")]
        else:
            cmd += ["-r", os.path.join(refio.GOLDEN, kind + ".fastq")]
        subprocess.run(cmd, check=True, cwd=wd, stderr=subprocess.DEVNULL)
        assert sorted(os.listdir(wd)) == sorted(info["files"])
        for fn in info["files"]:
            with open(os.path.join(wd, fn)) as f:
                assert f.read() == refio.golden_file(case, fn), "%s/%s" % (case, fn)


def test_overlong_read_is_refused(pat):
    import walt_amd
    p, _, _, idx, _ = pat
    limit = {5: 148, 7: 152}[p]
    ok = "ACGT" * 40
    idx.map_se_batch(*walt_amd.pack_reads([ok[:limit]]))
    with pytest.raises(walt_amd.WaltError) as ei:
        idx.map_se_batch(*walt_amd.pack_reads([ok[:limit + 1]]))
    assert "seed pattern %d" % p in str(ei.value)
