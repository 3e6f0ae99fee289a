"""Paired-end literal round in one launch (map_pe.hip k_pe_stage, option pe_lit_fuse): when the list of reads with a
truly dangerous probe is short enough for six work items per read to fit the queue, the launch of seed 0 takes all
three seed shifts (every probe made: a superset, k_pe_push applies the reference's exits, paired.cpp:133-149) and the
verifier runs once over the items of all seeds.  Whether a pass ran that way is control word 3 of each mate's block in
the workspace.  The pair records must be the oracle's either way: fused (default), seed by seed (pe_lit_fuse = 0), and
with a staged capacity too small for the fused form (pe_stage_cap: decided on the device from the list's length)."""
import os
import random

import numpy as np
import pytest

import refio

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wa():
    import walt_amd
    assert walt_amd.device_count() >= 1, "no HIP device: the walt_amd hot path has no CPU fallback"
    return walt_amd


def _genome(rng):
    """One long repeat-free part (most reads: no chromosome end shares their seed characters), and a family of 120
    short chromosomes cut from a second unit that also stands three times in a long sequence: a read from that unit
    meets chromosome-end entries (dangerous probes -> literal list) AND has dozens of candidates (regions of more than
    16 slots -> work items in the literal round)."""
    quiet = "".join(rng.choice("ACGT") for _ in range(6000))
    unit = "".join(rng.choice("ACGT") for _ in range(500))
    seqs = [("quiet", quiet), ("fam_long", unit * 3)]
    for i in range(120):
        a = rng.randrange(0, 250)
        L = rng.choice([70, 90, 131, 150, 200, 250])
        s_ = list(unit[a:a + L])
        for _ in range(rng.randrange(0, 3)):
            s_[rng.randrange(len(s_))] = rng.choice("ACGT")
        seqs.append(("f%d" % i, "".join(s_)))
    return seqs, quiet, unit * 3


def _pairs(rng, quiet, fam, n, fam_share):
    s1, s2 = [], []
    for _ in range(n):
        src = fam if rng.random() < fam_share else quiet
        flen = rng.randrange(120, 400)
        a = rng.randrange(0, len(src) - flen)
        frag = src[a:a + flen]
        if rng.random() < 0.5:
            frag = refio.revcomp(frag)
        frag = "".join("T" if (c == "C" and rng.random() < 0.9) else c for c in frag)
        L = rng.choice([60, 75, 100])
        s1.append(frag[:L])
        s2.append(refio.revcomp(frag)[:L])
    return s1, s2


def _device_call(wa, idx, s1, s2, top_k, max_mm, b, frag_range):
    import torch
    dev = torch.device("cuda:0")
    n = len(s1)
    b1, o1 = wa.pack_reads(s1)
    b2, o2 = wa.pack_reads(s2)
    d1, d2 = torch.from_numpy(b1).to(dev), torch.from_numpy(b2).to(dev)
    do1, do2 = torch.from_numpy(o1.astype(np.int64)).to(dev), torch.from_numpy(o2.astype(np.int64)).to(dev)
    d_out = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    d_stats = torch.zeros(8, dtype=torch.int64, device=dev)
    d_ws = torch.zeros(idx.pe_workspace_bytes(n, 100, top_k), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    idx.map_pe_batch_device(d1.data_ptr(), do1.data_ptr(), d2.data_ptr(), do2.data_ptr(), n, 100, d_out.data_ptr(),
                            d_stats.data_ptr(), d_ws.data_ptr(), d_ws.numel(), stream=stream, max_mismatches=max_mm, b=b,
                            top_k=top_k, frag_range=frag_range)
    idx.check_batch(d_ws.data_ptr(), stream)
    ctl = d_ws[:192 * 4].view(torch.int32).cpu().numpy()  # map_pe.hip carve_pe: [64 + 32 m ...] control block of mate m
    out = np.frombuffer(d_out.cpu().numpy().tobytes(), dtype=wa.pair_result_dtype)
    return out, {"literal": (int(ctl[64]), int(ctl[96])), "fused": (int(ctl[64 + 3]), int(ctl[96 + 3]))}


@pytest.mark.parametrize("top_k,max_mm,b", [(50, 6, 5000), (3, 6, 5000), (300, 10, 30)])
def test_gpu_pe_literal_round_in_one_launch_equals_seed_by_seed_and_oracle(wa, scratch, top_k, max_mm, b):
    rng = random.Random(20261005)
    seqs, quiet, fam = _genome(rng)
    fa = os.path.join(scratch, "pe_lit_fused.fa")
    with open(fa, "w") as f:
        for nm, s_ in seqs:
            f.write(">%s\n%s\n" % (nm, s_))
    path = os.path.join(scratch, "pe_lit_fused.dbindex")
    wa.makedb(fa, path, threads=4)
    db = refio.DbIndex(path)
    s1, s2 = _pairs(rng, quiet, fam, 3000, 0.12)
    want, _, _ = refio.oracle_pe(db, s1, s2, max_mm=max_mm, b=b, top_k=top_k, frag_range=1000)
    idx = wa.Index.open(path, device=0)
    try:
        runs = {}
        for name, opts in (("fused", {}), ("seed_by_seed", {"pe_lit_fuse": 0}), ("small_cap", {"pe_stage_cap": 256})):
            idx.set_option("pe_lit_fuse", 1)
            idx.set_option("pe_stage_cap", 0)
            for k, v in opts.items():
                idx.set_option(k, v)
            runs[name] = _device_call(wa, idx, s1, s2, top_k, max_mm, b, 1000)
        lit = runs["fused"][1]["literal"]
        assert min(lit) > 30, "the read set should send reads to the literal list: %r" % (lit,)
        assert 3 * max(lit) <= 3000, "the literal list should be short enough for the fused form: %r" % (lit,)
        assert runs["fused"][1]["fused"] == (1, 1), runs["fused"][1]
        assert runs["seed_by_seed"][1]["fused"] == (0, 0), runs["seed_by_seed"][1]
        assert runs["small_cap"][1]["fused"] == (0, 0), "3 x the list exceeds 256 staged reads: %r" % (runs["small_cap"][1],)
        for name, (out, _) in runs.items():
            for f in ("best_times", "frag_len", "pair_mm", "best_i", "best_j"):
                assert np.array_equal(out[f], want[f]), (name, f)
            for mate in ("m1", "m2"):
                for f in ("genome_pos", "times", "strand", "mismatch"):
                    assert np.array_equal(out[mate][f], want[mate][f]), (name, mate, f)
    finally:
        idx.set_option("pe_lit_fuse", 1)
        idx.set_option("pe_stage_cap", 0)
        idx.close()
