#!/usr/bin/env python3
"""End-to-end check at hg19 scale against the REAL reference binary (GPU box only).

  1. generate bench.py's hg19-scale synthetic genome on the GPU and build the index there,
  2. write it out in the reference's .dbindex format (walt_index_write) to a RAM-backed scratch dir,
  3. write a FASTQ sample of bench.py's synthetic reads,
  4. run walt_amd/bin/walt (GPU) and oracle/_ref/walt (the reference, -t <all cores>) on the same
     files with the same options, wall-clock both,
  5. compare the outputs byte for byte.

Prints one JSON line.  The reference binary is test infrastructure (oracle/_ref, built by
oracle/Makefile.ref from /root/reference in the build container); nothing here reads
/root/reference at run time.

  python3 tools/hg19_e2e.py --reads 4000000 [--mode se|pe] [--sam]
"""
import argparse
import filecmp
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def log(msg):
    print("[hg19_e2e] " + msg, file=sys.stderr, flush=True)


def pick_scratch(need_bytes):
    for base in (os.environ.get("WALT_AMD_SCRATCH"), "/dev/shm", "/tmp"):
        if not base or not os.path.isdir(base):
            continue
        if shutil.disk_usage(base).free > need_bytes * 1.2:
            d = os.path.join(base, "walt_amd_e2e_%d" % os.getpid())
            os.makedirs(d, exist_ok=True)
            return d
    raise SystemExit("no scratch directory with %.0f GB free" % (need_bytes / 1e9))


def write_fastq(path, bases, n, read_len, tag):
    """@<tag><9 digits>\\n<seq>\\n+\\n<qual>\\n per read, built with numpy (no per-read Python)."""
    rec = np.dtype([("at", "S1"), ("tag", "S1"), ("num", "S9"), ("nl0", "S1"), ("seq", "S%d" % read_len),
                    ("mid", "S3"), ("qual", "S%d" % read_len), ("nl1", "S1")])
    chunk = 1 << 20
    with open(path, "wb") as f:
        for s in range(0, n, chunk):
            m = min(chunk, n - s)
            a = np.zeros(m, dtype=rec)
            a["at"], a["tag"], a["nl0"], a["mid"], a["nl1"] = b"@", tag, b"\n", b"\n+\n", b"\n"
            a["num"] = np.char.zfill(np.arange(s, s + m).astype("S9"), 9)
            a["seq"] = bases[s * read_len:(s + m) * read_len].view("S%d" % read_len)
            a["qual"] = b"I" * read_len
            f.write(a.tobytes())


def run_timed(cmd):
    t0 = time.perf_counter()
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    dt = time.perf_counter() - t0
    if p.returncode != 0:
        raise SystemExit("command failed (%d): %s\n%s" % (p.returncode, " ".join(cmd), p.stdout[-2000:]))
    global LAST_STAGES
    LAST_STAGES = [ln for ln in p.stdout.splitlines() if ln.startswith("[walt_amd")]
    for ln in LAST_STAGES:
        log(ln)
    return dt


LAST_STAGES = []


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--small", type=int, default=0, help="second, smaller run to subtract the index load time")
    ap.add_argument("--genome-mbp", type=float, default=0.0, help="0 = full scale")
    ap.add_argument("--genome", choices=["hg19like", "easy"], default="hg19like")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--mode", choices=["se", "pe"], default="se")
    ap.add_argument("--sam", action="store_true")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--gpu-reads", type=int, default=0,
                    help="an additional run of bin/walt alone on this many reads (the reference would take minutes): end-to-end rate")
    ap.add_argument("--batch", type=int, default=10_000_000, help="-N of the additional run")
    args = ap.parse_args()

    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import synth
    import walt_amd
    import torch

    ref_bin = os.path.join(ROOT, "oracle", "_ref", "walt")
    our_bin = os.path.join(ROOT, "walt_amd", "bin", "walt")
    for b in (ref_bin, our_bin):
        if not os.path.exists(b):
            raise SystemExit("missing " + b + " (run __graft_entry__.build() in the build container)")
    threads = args.threads or walt_amd.effective_cpus()
    dev = torch.device("cuda", 0)
    full = synth.HG19_TOTAL if args.genome == 'hg19like' else sum(synth.HG19_CHROMS)
    scale = args.genome_mbp * 1e6 / full if args.genome_mbp else 1.0
    genome_ascii, lens, names = synth.make_genome(torch, dev, scale, seed=2, kind=args.genome)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    pe = args.mode == "pe"
    strands = walt_amd.STRANDS_ALL if pe else walt_amd.STRANDS_CT
    t0 = time.perf_counter()
    idx = walt_amd.Index.build_device(genome_ascii.data_ptr(), lens, names, device=0, strands=strands)
    t_build = time.perf_counter() - t0
    n_str = 4 if pe else 2
    need = n_str * (sum(lens) + 4 * (idx.index_size(0) + (1 << 24) + 8)) + 4 * args.reads * (2 * args.read_len + 20)
    scratch = pick_scratch(need)
    log("genome %d bp, index built in %.1f s, scratch %s (%.0f GB needed)" % (sum(lens), t_build, scratch, need / 1e9))
    out = {"genome_bp": int(sum(lens)), "mode": args.mode, "reads": args.reads, "read_len": args.read_len,
           "threads": threads, "index_build_s": round(t_build, 1), "format": "sam" if args.sam else "mr"}
    try:
        dbi = os.path.join(scratch, "hg.dbindex")
        t0 = time.perf_counter()
        idx.write(dbi)
        out["index_write_s"] = round(time.perf_counter() - t0, 1)
        log("index written in %.1f s" % out["index_write_s"])
        for sfx in ("_CT00", "_CT01", "_GA10", "_GA11"):  # both binaries only check that all four exist
            if not os.path.exists(dbi + sfx):
                open(dbi + sfx, "wb").close()
        if pe:
            b1, b2, _ = synth.make_pairs(torch, dev, genome_ascii, args.reads, args.read_len, seed=1000)
            h1, h2 = b1.cpu().numpy(), b2.cpu().numpy()
            del b1, b2
        else:
            b1, _ = synth.make_reads(torch, dev, genome_ascii, args.reads, args.read_len, seed=1000)
            h1 = b1.cpu().numpy()
            del b1
        idx.close()
        del genome_ascii
        torch.cuda.empty_cache()

        def files(n, tag):
            if pe:
                f1, f2 = os.path.join(scratch, "%s_1.fastq" % tag), os.path.join(scratch, "%s_2.fastq" % tag)
                write_fastq(f1, h1, n, args.read_len, b"p")
                write_fastq(f2, h2, n, args.read_len, b"p")
                return ["-1", f1, "-2", f2]
            f = os.path.join(scratch, "%s.fastq" % tag)
            write_fastq(f, h1, n, args.read_len, b"r")
            return ["-r", f]

        common = ["-i", dbi, "-m", "6", "-b", "5000", "-a", "-u", "-t", str(threads)]
        if pe:
            common += ["-k", "50", "-L", "1000"]
        if args.sam:
            common += ["-sam"]
        runs = [("full", args.reads)] + ([("small", args.small)] if args.small else [])
        for tag, n in runs:
            inp = files(n, tag)
            res = {}
            for who, binary in (("gpu", our_bin), ("ref", ref_bin)):
                o = os.path.join(scratch, "%s_%s.out" % (tag, who))
                dt = run_timed([binary] + inp + ["-o", o] + common + (["-v"] if who == "gpu" else []))
                res[who + "_wall_s"] = round(dt, 2)
                log("%s %s: %.1f s wall" % (tag, who, dt))
            sfx = [""] + ([] if args.sam else (["_1_ambiguous", "_1_unmapped", "_2_ambiguous", "_2_unmapped"] if pe
                                               else ["_ambiguous", "_unmapped"])) + [".mapstats"]
            same = {}
            for s in sfx:
                a, b = (os.path.join(scratch, "%s_%s.out%s" % (tag, w, s)) for w in ("gpu", "ref"))
                same[s or "main"] = os.path.exists(a) and os.path.exists(b) and filecmp.cmp(a, b, shallow=False)
            res["identical"] = same
            res["main_bytes"] = os.path.getsize(os.path.join(scratch, "%s_ref.out" % tag))
            res["mapstats"] = open(os.path.join(scratch, "%s_ref.out.mapstats" % tag)).read().split("\n")[:8]
            out[tag] = res
        if args.gpu_reads:
            # bin/walt alone on a larger input: several -N batches, so the loader's prefetch and the persistent device
            # buffers matter; rate = reads / (wall - index load), stage times from -v
            n_big = args.gpu_reads
            if pe:
                b1, b2, _ = synth.make_pairs(torch, dev, synth.make_genome(torch, dev, scale, seed=2, kind=args.genome)[0], n_big, args.read_len, seed=1001)
                h1, h2 = b1.cpu().numpy(), b2.cpu().numpy()
                del b1, b2
            else:
                b1, _ = synth.make_reads(torch, dev, synth.make_genome(torch, dev, scale, seed=2, kind=args.genome)[0], n_big, args.read_len, seed=1001)
                h1 = b1.cpu().numpy()
                del b1
            torch.cuda.empty_cache()
            inp = files(n_big, "big")
            o = os.path.join(scratch, "big_gpu.out")
            dt = run_timed([our_bin] + inp + ["-o", o] + common + ["-v", "-N", str(args.batch)])
            out["gpu_only"] = {"reads": n_big, "wall_s": round(dt, 2), "batch": args.batch, "stages": LAST_STAGES,
                               "output_bytes": os.path.getsize(o)}
        if args.small:
            dn = args.reads - args.small
            for who in ("gpu", "ref"):
                d = out["full"][who + "_wall_s"] - out["small"][who + "_wall_s"]
                out[who + "_reads_per_s_excl_index_load"] = dn / d if d > 0 else None
        out["all_identical"] = all(all(out[t]["identical"].values()) for t, _ in runs)
    finally:
        if not args.keep:
            shutil.rmtree(scratch, ignore_errors=True)
    print(json.dumps(out), flush=True)
    if not out.get("all_identical"):
        raise SystemExit(1)


if __name__ == "__main__":
    main()
