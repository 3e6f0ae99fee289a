#!/usr/bin/env python3
"""bench.py -- mapped reads/s of the single-end seed-and-extend hot path on
N x MI355X (BASELINE.json: metric "mapped reads/sec (100 bp, hg19)").

Workload at N=1 (BASELINE.json configs[1]): hg19-scale synthetic genome (24
chromosomes with hg19's lengths, 3.096 Gbp, iid bases + implanted repeat
families), its _CT00/_CT01 strand indexes built on the GPU by the product's
makedb-compatible builder, 50 M synthetic 100 bp single-end reads (both
strands, 95 % C->T, 1 % substitutions), -m 6 -b 5000.  A step = one pass of the
hot path over the whole resident batch (read packing + mapping kernels, both
strand passes).  For N > 1 every rank holds a full index replica and its own
50 M-read shard (weak scaling, BASELINE.json configs[3]); the only collective is
the final all-reduce of the mapping statistics over RCCL.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     : algorithmic bytes of the implemented search per launch (128-byte
                 line per dependent gather) / HIP-event duration
                 of the mapping kernels, vs 8 TB/s; measured HBM traffic beside it
  cpu_baseline : the oracle restatement (bit-exact to the reference, OpenMP on
                 all host cores) timed on a bounded sample of the same reads.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HG19 = [249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022, 141213431,
        135534747, 135006516, 133851895, 115169878, 107349540, 102531392, 90354753, 81195210, 78077248, 59128983,
        63025520, 48129895, 51304566, 155270560, 59373566]
HG19_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW"


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench] " + msg, file=sys.stderr, flush=True)


def make_genome(torch, dev, scale, seed, contigs=0):
    """ASCII genome on the GPU: iid bases + repeat families (deterministic per seed)."""
    lens = [max(1000, int(l * scale)) for l in HG19]
    if contigs:  # same bases cut into equal contigs: an assembly of many scaffolds (chromosome-end handling)
        total = sum(lens)
        lens = [total // contigs] * (contigs - 1) + [total - (total // contigs) * (contigs - 1)]
    L = sum(lens)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    codes = torch.empty(L, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for s in range(0, L, step):
        e = min(L, s + step)
        codes[s:e] = torch.randint(0, 4, (e - s,), generator=g, device=dev, dtype=torch.uint8)

    def implant(unit_len, copies, divergence):
        if copies < 1:
            return
        unit = torch.randint(0, 4, (unit_len,), generator=g, device=dev, dtype=torch.uint8)
        slot = max(unit_len + 64, L // (copies + 1))
        nslots = (L - unit_len - 64) // slot
        copies_eff = min(copies, nslots)
        which = torch.randperm(nslots, generator=g, device=dev)[:copies_eff]
        jitter = torch.randint(0, max(1, slot - unit_len - 32), (copies_eff,), generator=g, device=dev)
        starts = which * slot + jitter
        for c0 in range(0, copies_eff, 1 << 20):
            st = starts[c0:c0 + (1 << 20)]
            idx = (st[:, None] + torch.arange(unit_len, device=dev)[None, :]).reshape(-1)
            vals = unit.repeat(st.numel())
            if divergence > 0:
                mut = torch.rand(vals.numel(), generator=g, device=dev) < divergence
                rnd = torch.randint(1, 4, (vals.numel(),), generator=g, device=dev, dtype=torch.uint8)
                vals = torch.where(mut, (vals + rnd) & 3, vals)
            codes[idx] = vals

    implant(300, int(20000 * scale), 0.10)    # SINE-like family
    implant(6000, int(500 * scale), 0.02)     # LINE-like family
    implant(48, int(600000 * scale), 0.0)     # exact micro-repeat: raw bucket >= 500000 is erased (reference.cpp:211)
    implant(150, int(8000 * scale), 0.0)      # exact repeat: narrowed region > -b 5000 is skipped (mapping.cpp:275)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    ascii_g = lut[codes.long()] if L < (1 << 28) else torch.cat([lut[codes[s:s + step].long()] for s in range(0, L, step)])
    return ascii_g, lens


def make_reads(torch, dev, genome_ascii, n, read_len, seed):
    """n x read_len ASCII reads on the GPU: both strands, 95 % C->T, 1 % substitutions."""
    L = genome_ascii.numel()
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = torch.empty((n, read_len), dtype=torch.uint8, device=dev)
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in ((65, 84), (67, 71), (71, 67), (84, 65)):
        comp[a] = b
    ar = torch.arange(read_len, device=dev)
    chunk = 1 << 22
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        pos = torch.randint(0, L - read_len, (m,), generator=g, device=dev)
        r = genome_ascii[pos[:, None] + ar[None, :]]
        rev = torch.rand(m, generator=g, device=dev) < 0.5
        rc = comp[r.flip(1).long()]
        r = torch.where(rev[:, None], rc, r)
        conv = (r == 67) & (torch.rand(r.shape, generator=g, device=dev) < 0.95)
        r = torch.where(conv, torch.full_like(r, 84), r)
        sub = torch.rand(r.shape, generator=g, device=dev) < 0.01
        lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
        rnd = lut[torch.randint(0, 4, r.shape, generator=g, device=dev)]
        r = torch.where(sub, rnd, r)
        out[s:s + m] = r
    offsets = torch.arange(n + 1, device=dev, dtype=torch.int64) * read_len
    return out.reshape(-1), offsets


def make_pairs(torch, dev, genome_ascii, n, read_len, seed):
    """n pairs on the GPU: fragment length U[120,500] from either strand, bisulfite (95 % C->T on the
    fragment), mate 1 = fragment[:L], mate 2 = revcomp(fragment)[:L], 1 % substitutions."""
    L = genome_ascii.numel()
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    m1 = torch.empty((n, read_len), dtype=torch.uint8, device=dev)
    m2 = torch.empty((n, read_len), dtype=torch.uint8, device=dev)
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in ((65, 84), (67, 71), (71, 67), (84, 65)):
        comp[a] = b
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    ar = torch.arange(read_len, device=dev)
    chunk = 1 << 21
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        flen = torch.randint(max(120, read_len), 501, (m,), generator=g, device=dev)
        pos = torch.randint(0, L - 600, (m,), generator=g, device=dev)
        rev = torch.rand(m, generator=g, device=dev) < 0.5
        # 5' end of the fragment on its own strand, and the 5' end of the opposite strand
        left = genome_ascii[pos[:, None] + ar[None, :]]                                  # genome[pos : pos+L]
        right = comp[genome_ascii[(pos + flen)[:, None] - 1 - ar[None, :]].long()]       # revcomp of the last L bases
        f5 = torch.where(rev[:, None], right, left)     # fragment[:L]
        f3rc = torch.where(rev[:, None], left, right)   # what revcomp(fragment)[:L] is BEFORE conversion ...
        # bisulfite acts on the fragment strand: C->T in f5; the mate-2 read is the reverse complement of the
        # converted fragment end, i.e. G->A relative to the opposite strand
        r1 = torch.where((f5 == 67) & (torch.rand(f5.shape, generator=g, device=dev) < 0.95), torch.full_like(f5, 84), f5)
        r2 = torch.where((f3rc == 71) & (torch.rand(f5.shape, generator=g, device=dev) < 0.95), torch.full_like(f5, 65), f3rc)
        for r, dst in ((r1, m1), (r2, m2)):
            sub = torch.rand(r.shape, generator=g, device=dev) < 0.01
            rnd = lut[torch.randint(0, 4, r.shape, generator=g, device=dev)]
            dst[s:s + m] = torch.where(sub, rnd, r)
    offsets = torch.arange(n + 1, device=dev, dtype=torch.int64) * read_len
    return m1.reshape(-1), m2.reshape(-1), offsets


def run_pe(args, torch, walt_amd, dev, local, rank, world, genome_ascii, lens):
    """configs[2]: paired-end mapping (top-k heaps + pair merge on the device); reports pairs/s."""
    t0 = time.perf_counter()
    idx = walt_amd.Index.build_device(genome_ascii.data_ptr(), lens, HG19_NAMES, device=local,
                                      strands=walt_amd.STRANDS_ALL, dir_bits=args.dir_bits)
    t_index = time.perf_counter() - t0
    log("index (4 strands): %.1f GB in HBM, dir_bits %d (%.1f s)" % (idx.device_bytes / 1e9, idx.dir_bits, t_index))
    n = args.reads
    d1, d2, d_off = make_pairs(torch, dev, genome_ascii, n, args.read_len, seed=2000 + rank)
    torch.cuda.synchronize()
    run_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    ns = min(args.cpu_sample // 4, n)  # paired-end costs ~3.4x single-end per read on the CPU
    if run_cpu:
        m1_host, m2_host = d1[:ns * args.read_len].cpu().numpy(), d2[:ns * args.read_len].cpu().numpy()
    del genome_ascii
    torch.cuda.empty_cache()
    d_out = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    d_stats = torch.zeros(8, dtype=torch.int64, device=dev)
    d_ws = torch.empty(walt_amd.lib().walt_pe_workspace_bytes(n, args.read_len, args.top_k), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        rc = walt_amd.lib().walt_map_pe_batch_device(idx.handle, d1.data_ptr(), d_off.data_ptr(), d2.data_ptr(),
                                                     d_off.data_ptr(), n, args.read_len, args.max_mismatches,
                                                     args.bucket, args.top_k, args.frag_range, d_out.data_ptr(),
                                                     d_stats.data_ptr(), d_ws.data_ptr(), stream)
        assert rc == 0, walt_amd.lib().walt_last_error()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    idx.check_batch(d_ws.data_ptr(), stream)  # no invalid read went unnoticed
    from walt_amd import dist as wdist
    elapsed = wdist.allreduce_max(elapsed, device=dev)  # MAX over ranks
    ctl = d_ws[:192 * 4].view(torch.int32).cpu().numpy()  # control words of the last chunk (map_pe.hip carve_pe)
    log("last chunk: literal-list %d / %d, of which overflowed the 8-slot heaps %d / %d (mate 1 / mate 2), heavy pairs %d" % (
        int(ctl[64]), int(ctl[96]), int(ctl[89]), int(ctl[121]), int(ctl[128])))
    res = d_out.view(torch.int32).view(n, 16)
    bt = res[:, 8]
    # StatPairedReads pair counters (paired.hpp:96-105), summed over ranks: the only collective
    pst = wdist.allreduce_stats(torch.stack([torch.tensor(n, device=dev), (bt == 1).sum(), (bt >= 2).sum(),
                                             (bt == 0).sum()]).to(torch.int64))
    pairs_t, uniq_t, amb_t, unp_t = [int(v) for v in pst.tolist()]
    out = {"metric": "mapped read pairs/sec (2 x %d bp paired-end, hg19-scale index, -m %d -k %d -L %d)" % (
               args.read_len, args.max_mismatches, args.top_k, args.frag_range),
           "value": world * n * args.steps / elapsed, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
           "config": {"workload": "configs[2]: hg19-scale synthetic genome, %d pairs 2 x %d bp, fragment U[120,500]" % (
               n, args.read_len), "index_hbm_gb": round(idx.device_bytes / 1e9, 2)},
           "mapping": {"pairs": pairs_t, "unique_pairs": uniq_t, "ambiguous_pairs": amb_t, "unpaired": unp_t}}
    if run_cpu:
        want, cpu_s, cores, works = cpu_baseline_pe(idx, m1_host, m2_host, args.read_len, ns, lens, args.max_mismatches,
                                                    args.bucket, args.top_k, args.frag_range)
        got = d_out[:ns * 64].cpu().numpy().view(walt_amd.pair_result_dtype)
        same = all(np.array_equal(got[f], want[f]) for f in ("best_times", "frag_len", "best_i", "best_j", "pair_mm"))
        for m in ("m1", "m2"):
            same = same and all(np.array_equal(got[m][f], want[m][f]) for f in ("genome_pos", "times", "strand", "mismatch"))
        out["cpu_baseline"] = {"value": ns / cpu_s, "unit": "pairs/s", "cores": cores, "kind": "port",
                               "sample": "first %d pairs, oracle restatement of PairEndMapping on both mates and "
                                         "strands + pair merge, OpenMP; two strand indexes in host memory at a time" % ns,
                               "bit_exact_vs_gpu": bool(same)}
        # Same accounting as the single-end line: per pair, P probes and C candidates over
        # both mates (oracle counters) -> 2 P + C dependent gathers of one 128-byte line each, plus per mate the
        # ranked list written by the top-k kernel and read back by the merge (one line each way), the packed
        # reads and the 64-byte pair record.  Time = the whole step (the mates' kernels overlap on several
        # streams, so no single kernel's duration is meaningful); traffic from profiles/traffic_pe.json when it matches.
        P = sum(w[0] for w in works) / ns
        C = sum(w[1] for w in works) / ns
        bytes_per_pair = 128.0 * (2.0 * P + C + 4.0) + 2 * args.read_len / 4.0 + 64
        step_s = elapsed / args.steps
        traffic = None  # PMC figure of tools/prof_pmc.sh ... --mode pe + tools/pe_traffic.py, for this exact workload
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "traffic_pe.json")))
            if (t.get("pairs_per_step") == n and args.read_len == 100 and args.max_mismatches == 6 and
                    args.bucket == 5000 and args.top_k == 50 and args.pattern == 3 and not args.contigs):
                traffic = t["hbm_bytes_per_step"]
        except (OSError, ValueError, KeyError):
            pass
        out["roofline"] = {"bound": "hbm", "achieved": bytes_per_pair * n / step_s / 1e9, "peak": HBM_PEAK / 1e9,
                           "unit": "GB/s", "frac": bytes_per_pair * n / step_s / HBM_PEAK, "traffic": traffic,
                           "traffic_frac": (traffic / step_s / HBM_PEAK) if traffic else None,
                           "kernel": "whole paired-end step (k_pe_topk_dual + list kernels of both mates, k_pe_merge)",
                           "algorithmic_bytes_per_pair": bytes_per_pair,
                           "granularity": "128-byte line per dependent gather (2 per probe, 1 per candidate, 2 per "
                                          "mate for the ranked list) + streamed reads / pair record",
                           "per_pair": {"probes": P, "candidates": C}}
    if rank == 0:
        print(json.dumps(out), flush=True)
    idx.close()


def cpu_baseline(idx, reads_host, read_len, n_sample, lens, max_mm, b):
    """Oracle restatement on the host cores, one strand index in memory at a time
    (as the reference itself does, mapping.cpp:491-492)."""
    import refio
    import walt_amd
    cores = walt_amd.effective_cpus()  # affinity capped by the cgroup CPU quota (the GPU box grants 16 of 256)
    orc = refio.oracle()
    n = n_sample
    bases = np.ascontiguousarray(reads_host[:n * read_len])
    offsets = (np.arange(n + 1, dtype=np.uint64) * read_len)
    start = np.zeros(len(lens) + 1, dtype=np.uint32)
    start[1:] = np.cumsum(lens, dtype=np.uint64).astype(np.uint32)
    out = np.zeros(n, dtype=refio.best_dtype)
    work = np.zeros(1, dtype=refio.work_dtype)
    orc.orc_se_init(out.ctypes.data, n, max_mm)
    elapsed = 0.0
    for strand, ch in ((0, b"+"), (1, b"-")):
        g, cnt, ix = idx.export_strand(strand)
        x = refio.make_orc_strand(g, cnt, ix, start)
        t0 = time.perf_counter()
        orc.orc_se_map_strand(ctypes.addressof(x), ch, bases.ctypes.data, offsets.ctypes.data, n, 0, b, cores,
                              out.ctypes.data, work.ctypes.data)
        elapsed += time.perf_counter() - t0
        del g, cnt, ix, x
    return out, work[0], elapsed, cores


def reference_binary_leg(idx, d_bases, d_out, read_len, n_ref, max_mm, b, cores):
    """The REAL reference binary (oracle/_ref/walt, built from the reference sources by oracle/Makefile.ref;
    test infrastructure) on the same box in the same run: the resident index is written in the reference's
    .dbindex format to a RAM-backed scratch directory, the first n_ref reads of the batch go to a FASTQ file,
    `walt -t <cores>` maps them, and its .mapstats is compared with the GPU's records for the same reads.
    Returns a dict (or a dict with "skipped")."""
    import shutil
    import subprocess
    import tempfile

    ref_bin = os.path.join(ROOT, "oracle", "_ref", "walt")
    if not os.path.exists(ref_bin):
        return {"skipped": "oracle/_ref/walt is not built"}
    need = 2 * (idx.genome_len + 4 * (idx.index_size(0) + (1 << 24) + 8)) + n_ref * (2 * read_len + 20) * 2
    base = None
    for cand in (os.environ.get("WALT_AMD_SCRATCH"), "/dev/shm", "/tmp"):
        if cand and os.path.isdir(cand) and shutil.disk_usage(cand).free > 1.3 * need:
            base = cand
            break
    try:
        avail_kb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0]
    except (OSError, IndexError, ValueError):
        avail_kb = 0
    if base is None or avail_kb * 1024 < 2.2 * need:
        return {"skipped": "no scratch space / host memory for a %.0f GB index copy" % (need / 1e9)}
    scratch = tempfile.mkdtemp(prefix="walt_amd_bench_ref_", dir=base)
    try:
        dbi = os.path.join(scratch, "hg.dbindex")
        t0 = time.perf_counter()
        idx.write(dbi)
        for sfx in ("_GA10", "_GA11"):  # the binary only checks that all four strand files exist
            if not os.path.exists(dbi + sfx):
                open(dbi + sfx, "wb").close()
        t_write = time.perf_counter() - t0
        host = d_bases[:n_ref * read_len].cpu().numpy()
        fq = os.path.join(scratch, "sample.fastq")
        rec = np.dtype([("at", "S2"), ("num", "S9"), ("nl0", "S1"), ("seq", "S%d" % read_len), ("mid", "S3"),
                        ("qual", "S%d" % read_len), ("nl1", "S1")])
        with open(fq, "wb") as f:
            for s0 in range(0, n_ref, 1 << 20):
                m = min(1 << 20, n_ref - s0)
                a = np.zeros(m, dtype=rec)
                a["at"], a["nl0"], a["mid"], a["nl1"] = b"@r", b"\n", b"\n+\n", b"\n"
                a["num"] = np.char.zfill(np.arange(s0, s0 + m).astype("S9"), 9)
                a["seq"] = host[s0 * read_len:(s0 + m) * read_len].view("S%d" % read_len)
                a["qual"] = b"I" * read_len
                f.write(a.tobytes())
        out = os.path.join(scratch, "ref.mr")
        cmd = [ref_bin, "-i", dbi, "-r", fq, "-o", out, "-m", str(max_mm), "-b", str(b), "-t", str(cores)]
        t0 = time.perf_counter()
        pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        wall = time.perf_counter() - t0
        if pr.returncode != 0:
            return {"skipped": "reference binary failed: " + pr.stdout[-300:]}
        stats = {}
        for ln in open(out + ".mapstats"):
            k, _, v = ln.strip().partition(":")
            if v.strip().lstrip("-").replace(".", "", 1).isdigit():
                stats[k.strip()] = v.strip()
        times = d_out[:n_ref * 16].view(torch_int32()).view(n_ref, 4)[:, 1]
        mine = {"unique": int((times == 1).sum().item()), "ambiguous": int((times >= 2).sum().item()),
                "unmapped": int((times == 0).sum().item())}
        same = all(int(stats.get(k, -1)) == v for k, v in mine.items())
        return {"value": n_ref / wall, "unit": "reads/s", "cores": cores, "kind": "reference",
                "sample": "oracle/_ref/walt -t %d on the first %d reads of the batch, wall clock of the whole run "
                          "including its read of both strand index files (%.0f GB, RAM-backed) -- the reference "
                          "reloads them for every batch (mapping.cpp:491-492)" % (
                              cores, n_ref, 2 * (idx.genome_len + 4 * (idx.index_size(0) + (1 << 24) + 8)) / 1e9),
                "wall_s": wall, "index_write_s": t_write, "mapstats_equal_gpu": bool(same)}
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


def torch_int32():
    import torch
    return torch.int32


def cpu_baseline_pe(idx, m1_host, m2_host, read_len, n, lens, max_mm, b, top_k, frag_range):
    """Oracle restatement of PairEndMapping + MergePairedEndResults on the host cores; the two strand
    indexes of one mate are in host memory at a time."""
    import refio
    import walt_amd
    cores = walt_amd.effective_cpus()
    orc = refio.oracle()
    offsets = (np.arange(n + 1, dtype=np.uint64) * read_len)
    start = np.zeros(len(lens) + 1, dtype=np.uint32)
    start[1:] = np.cumsum(lens, dtype=np.uint64).astype(np.uint32)
    ranked, counts = [], []
    works = []
    elapsed = 0.0
    for mate, bases in ((0, m1_host), (1, m2_host)):
        keep = []
        arr = (refio.OrcStrand * 2)()
        for k in range(2):
            g, cnt, ix = idx.export_strand(2 * mate + k)
            keep.append((g, cnt, ix))
            x = refio.make_orc_strand(g, cnt, ix, start)
            for f, _ in refio.OrcStrand._fields_:
                setattr(arr[k], f, getattr(x, f))
        r = np.zeros((n, top_k), dtype=refio.cand_dtype)
        c = np.zeros(n, dtype=np.uint32)
        work = np.zeros(1, dtype=refio.work_dtype)
        bases = np.ascontiguousarray(bases[:n * read_len])
        t0 = time.perf_counter()
        orc.orc_pe_topk_batch(ctypes.addressof(arr), bases.ctypes.data, offsets.ctypes.data, n, mate, max_mm, b, top_k,
                              cores, r.ctypes.data, c.ctypes.data, work.ctypes.data)
        elapsed += time.perf_counter() - t0
        ranked.append(r)
        counts.append(c)
        works.append((float(work[0]["probes"]), float(work[0]["cands"])))
        del keep, arr
    out = np.zeros(n, dtype=refio.pair_dtype)
    t0 = time.perf_counter()
    orc.orc_pe_merge_batch(ranked[0].ctypes.data, counts[0].ctypes.data, ranked[1].ctypes.data, counts[1].ctypes.data,
                           top_k, offsets.ctypes.data, offsets.ctypes.data, n, start.ctypes.data, len(lens), frag_range,
                           max_mm, out.ctypes.data)
    elapsed += time.perf_counter() - t0
    return out, elapsed, cores, works


def main():
    # the paired-end path keeps several streams busy (two mates x two pipeline slots); the HIP runtime
    # multiplexes streams onto 4 hardware queues unless told otherwise, before it initialises
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=3095.677412, help="synthetic genome size (hg19 = 3095.68)")
    ap.add_argument("--reads", type=int, default=50_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--cpu-sample", type=int, default=400_000)
    ap.add_argument("--max-mismatches", type=int, default=6)
    ap.add_argument("--bucket", type=int, default=5000)
    ap.add_argument("--dir-bits", type=int, default=-1)
    ap.add_argument("--contigs", type=int, default=0, help="cut the genome into this many equal contigs (default: hg19's 24)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["se", "pe"], default="se", help="se = configs[1] (headline), pe = configs[2]")
    ap.add_argument("--top-k", type=int, default=50)
    ap.add_argument("--frag-range", type=int, default=1000)
    ap.add_argument("--ref-sample", type=int, default=3_000_000,
                    help="reads the real reference binary (oracle/_ref/walt) maps beside the oracle port at N=1 "
                         "(0 = skip; needs ~35 GB of RAM-backed scratch for the index copy)")
    ap.add_argument("--slot-table", action="store_true",
                    help="build the opt-in direct-mapped slot table (WALT_AMD_TABLE=1: +51.5 GB per strand at hg19 scale)")
    ap.add_argument("--pattern", type=int, choices=[3, 5, 7], default=3,
                    help="seed pattern (the reference's -D SEEDPATTERN3/5/7); 5 and 7 use libwalt_amd_sp5/_sp7.so, "
                         "whose kernels search literally -- not the headline configuration")
    args = ap.parse_args()

    if args.slot_table:
        os.environ["WALT_AMD_TABLE"] = "1"
    import walt_amd  # loads the HIP library (and the HIP runtime torch will share)
    import refio
    walt_amd.set_pattern(args.pattern)
    refio.set_pattern(args.pattern)  # the oracle build of the cpu_baseline leg
    walt_amd.lib()
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("WALT_AMD_BENCH_SHARE_GPU"):
            # rehearsal of the multi-rank path on a one-GPU box: every rank on GPU 0, gloo instead of RCCL
            # (RCCL refuses two ranks on one device); not a measurement
            local = 0
            torch.cuda.set_device(local)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if walt_amd.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the walt_amd hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    scale = args.genome_mbp * 1e6 / sum(HG19)
    t0 = time.perf_counter()
    genome_ascii, lens = make_genome(torch, dev, scale, seed=2, contigs=args.contigs)
    names = HG19_NAMES if not args.contigs else ["ctg%d" % i for i in range(len(lens))]
    torch.cuda.synchronize()
    t_genome = time.perf_counter() - t0
    log("genome: %d bp in %d chromosomes (%.1f s)" % (sum(lens), len(lens), t_genome))

    torch.cuda.empty_cache()  # hand the generator's cached blocks back: the library allocates with hipMalloc
    if args.mode == "pe":
        run_pe(args, torch, walt_amd, dev, local, rank, world, genome_ascii, lens)
        if world > 1:
            dist.destroy_process_group()
        return
    t0 = time.perf_counter()
    idx = walt_amd.Index.build_device(genome_ascii.data_ptr(), lens, names, device=local,
                                      strands=walt_amd.STRANDS_CT, dir_bits=args.dir_bits)
    t_index = time.perf_counter() - t0
    log("index: CT00 %d + CT01 %d entries, dir_bits %d, %.1f GB in HBM, outliers %d/%d, bad buckets %d/%d (%.1f s)" % (
        idx.index_size(0), idx.index_size(1), idx.dir_bits, idx.device_bytes / 1e9, idx.outliers(0),
        idx.outliers(1), idx.bad_buckets(0), idx.bad_buckets(1), t_index))

    n = args.reads
    t0 = time.perf_counter()
    d_bases, d_off = make_reads(torch, dev, genome_ascii, n, args.read_len, seed=1000 + rank)
    torch.cuda.synchronize()
    log("reads: %d x %d bp (%.1f s)" % (n, args.read_len, time.perf_counter() - t0))
    reads_sample_host = None
    run_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline  # the CPU leg runs at N=1 only
    if run_cpu:
        reads_sample_host = d_bases[:args.cpu_sample * args.read_len].cpu().numpy()
    del genome_ascii
    torch.cuda.empty_cache()

    d_out = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    d_stats = torch.zeros(4, dtype=torch.int64, device=dev)
    d_ws = torch.empty(walt_amd.lib().walt_se_workspace_bytes(n, args.read_len), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    idx.profile_enable(True)

    def step():
        idx.map_se_batch_device(d_bases.data_ptr(), d_off.data_ptr(), n, args.read_len, d_out.data_ptr(),
                                d_stats.data_ptr(), d_ws.data_ptr(), stream=stream,
                                max_mismatches=args.max_mismatches, b=args.bucket)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    pack_ms, map_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        p_ms, m_ms = idx.profile_last()  # waits for this step's events (same stream)
        pack_ms.append(p_ms)
        map_ms.append(m_ms)
    barrier()
    elapsed = time.perf_counter() - t0
    idx.check_batch(d_ws.data_ptr(), stream)  # no invalid read went unnoticed
    from walt_amd import dist as wdist
    elapsed = wdist.allreduce_max(elapsed, device=dev)  # MAX over ranks

    # mapping statistics of the last step; the ONLY data-path collective is this
    # final sum over ranks (RCCL), mirroring StatSingleReads (mapping.hpp:94-100)
    times = d_out.view(torch.int32).view(n, 4)[:, 1]
    too_short = int(d_stats[0].item()) // (args.warmup + args.steps)
    st = wdist.allreduce_stats(wdist.se_stats_vector(times, too_short))
    total, uniq, amb, unm, short = [int(v) for v in st.tolist()]

    if os.environ.get("WALT_AMD_STAMPS"):
        buf = (ctypes.c_ulonglong * 16)()
        if walt_amd.lib().walt_profile_stamps(buf) == 0:
            tot = float(buf[8]) or 1.0
            names = ["read record", "care loads", "bloom/bad", "lookup", "masks", "own-lane verify", "coop regions",
                     "store", "total"]
            log("phase shares (s_memtime, drained at boundaries): " +
                ", ".join("%s %.1f%%" % (nm, 100.0 * buf[i] / tot) for i, nm in enumerate(names[:8])))
            if int(os.environ.get("WALT_AMD_ABLATE", "0")) & 8:
                log("danger-filter self-check: %d probe pairs checked, %d dangerous probes NOT flagged by the filter "
                    "(must be 0)" % (buf[14], buf[15]))
    ctl = d_ws[:64 * 4].view(torch.int32).cpu().numpy()
    log("deferred to the literal pass: %d reads (bins %s)" % (int(ctl[32]), ctl[40:46].tolist()))
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * n * args.steps / elapsed
    if rank == 0:
        out = {
            "metric": "mapped reads/sec (%d bp single-end, hg19-scale index, -m %d -b %d)%s" % (
                args.read_len, args.max_mismatches, args.bucket,
                "" if args.pattern == 3 else ", seed pattern %d" % args.pattern),
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "configs[1]: hg19-scale synthetic genome (%d bp, 24 chromosomes, repeat families), "
                                   "%d x %d bp single-end C->T reads per GPU, -m %d -b %d" % (
                                       sum(lens), n, args.read_len, args.max_mismatches, args.bucket),
                       "genome_bp": int(sum(lens)), "reads_per_gpu": n, "read_len": args.read_len,
                       "max_mismatches": args.max_mismatches, "bucket_cap": args.bucket,
                       "index_dir_bits": idx.dir_bits, "index_hbm_gb": round(idx.device_bytes / 1e9, 2),
                       "index_build_s": round(t_index, 1), "parallelism": "replica-per-gpu x%d" % world},
            "mapping": {"total": total, "unique": uniq, "ambiguous": amb, "unmapped": unm, "too_short": short},
            "kernel_ms": {"pack_reads": float(np.mean(pack_ms)), "map_se": float(np.mean(map_ms))},
        }
        # HBM bytes per launch from the PMC passes of tools/prof_pmc.sh on this same command
        # (TCC_EA0_RDREQ x 128 B + WRITE_SIZE; profiles/traffic.json), null when not collected
        traffic, stored = None, None
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj):
            try:
                t = json.load(open(tj))
                # the stored PMC figure belongs to the headline workload only
                if (t.get("reads_per_launch") == n and t.get("genome_bp") == int(sum(lens)) and args.read_len == 100
                        and args.max_mismatches == 6 and args.bucket == 5000 and args.pattern == 3
                        and not args.contigs and not os.environ.get("WALT_AMD_TABLE")):
                    traffic = t["hbm_bytes_per_launch"]
                    stored = t.get("algorithmic_per_read")
            except (ValueError, KeyError):
                pass
        per_read, per_read_src = None, None
        if run_cpu:
            ns = min(args.cpu_sample, n)
            ref, work, cpu_s, cores = cpu_baseline(idx, reads_sample_host, args.read_len, ns, lens,
                                                   args.max_mismatches, args.bucket)
            got = d_out[:ns * 16].cpu().numpy().view(walt_amd.best_match_dtype)
            same = all(np.array_equal(got[f], ref[f]) for f in ("genome_pos", "times", "strand", "mismatch"))
            per_read = {"probes": float(work["probes"]) / ns, "search_steps": float(work["steps"]) / ns,
                        "candidates": float(work["cands"]) / ns}
            per_read_src = "oracle counters on this run's sample"
            out["cpu_baseline"] = {"value": ns / cpu_s, "unit": "reads/s", "cores": cores, "kind": "port",
                                   "sample": "first %d reads of rank 0's batch, both strand passes, oracle "
                                             "restatement with OpenMP; index in host memory" % ns,
                                   "bit_exact_vs_gpu": bool(same)}
            if args.ref_sample > 0 and args.pattern == 3:
                # the real reference binary beside it (slower than the port: it also reads its index files)
                try:
                    out["cpu_baseline"]["reference_binary"] = reference_binary_leg(
                        idx, d_bases, d_out, args.read_len, min(args.ref_sample, n), args.max_mismatches,
                        args.bucket, cores)
                except Exception as e:  # never let the extra leg break the bench line
                    out["cpu_baseline"]["reference_binary"] = {"skipped": "%s: %s" % (type(e).__name__, e)}
        elif stored and args.read_len == 100 and args.max_mismatches == 6 and args.bucket == 5000:
            per_read = {k: float(stored[k]) for k in ("probes", "search_steps", "candidates")}
            per_read_src = "profiles/traffic.json (oracle counters of the N=1 run of this workload)"
        if per_read:
            # Algorithmic bytes per read of the IMPLEMENTED search.  P probes and C verified
            # candidates are the reference algorithm's own counts (oracle); per probe the
            # directory/key search must read one directory pair and one run of entries, per candidate one
            # genome window -- dependent random gathers, which HBM serves in whole 128-byte lines (every
            # TCC_EA0_RDREQ of this kernel is a 128-byte request, profiles/): 2 P + C lines, plus the packed
            # read (L/4 bytes) and the 16-byte result, which stream.  This is a lower bound of the traffic
            # (measured: `traffic`), unlike SURVEY 8(d)'s formula, which prices the REFERENCE algorithm's
            # ~540 binary-search steps per read and is kept below as `survey_8d` (it exceeds the peak because
            # the directory replaces those steps).
            P, S, C = per_read["probes"], per_read["search_steps"], per_read["candidates"]
            table = bool(os.environ.get("WALT_AMD_TABLE", "0") not in ("", "0"))
            lines = (1.0 if table else 2.0) * P + C  # with the slot table a probe can be served by one line
            bytes_per_read = 128.0 * lines + args.read_len / 4.0 + 16
            useful = args.read_len / 4.0 + 16 + P * (8 + 12) + C * 32.0  # the same accesses counted in useful bytes
            kern_s = float(np.mean(map_ms)) / 1e3
            achieved = bytes_per_read * n / kern_s  # this rank's kernel: bytes of ITS launch / ITS duration
            survey = args.read_len + 16 + 8 * P + S * 4.25 + C * (4 + args.read_len / 4.0)
            out["roofline"] = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK, "traffic": traffic,
                               "traffic_frac": (traffic / kern_s / HBM_PEAK) if traffic else None,
                               "kernel": "k_map_se<7> (+ literal pass)" if args.pattern == 3 else "k_map_se_literal<7> (every read; literal search)",
                               "algorithmic_bytes_per_read": bytes_per_read,
                               "granularity": "128-byte line per dependent gather (%d per probe, 1 per candidate) + streamed read/result bytes" % (1 if table else 2),
                               "useful_bytes_per_read": useful, "per_read": per_read,
                               "per_read_source": per_read_src,
                               "survey_8d": {"bytes_per_read": survey, "achieved": survey * n / kern_s / 1e9,
                                             "frac": survey * n / kern_s / HBM_PEAK,
                                             "note": "bytes of the reference algorithm (binary-search steps S); not a "
                                                     "bound on this kernel, which replaces them by a directory lookup"}}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()  # ranks > 0 wait for rank 0's CPU baseline before tearing the group down
    idx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
