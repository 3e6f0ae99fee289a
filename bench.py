#!/usr/bin/env python3
"""bench.py -- mapped reads/s of the single-end seed-and-extend hot path on
N x MI355X (BASELINE.json: metric "mapped reads/sec (100 bp, hg19)").

Workload at N=1 (BASELINE.json configs[1]): an hg19-LIKE synthetic genome (tools/synth.py: hg19's 93
sequences, 3,137,161,264 bp, Alu-/L1-like families, segmental duplications, satellites, simple repeats --
calibrated against the reference's own tables of region sizes and unique fractions on real hg19), its
_CT00/_CT01 strand indexes built on the GPU by the product's makedb-compatible builder, 50 M synthetic
100 bp single-end reads (both strands, 95 % C->T, 1 % substitutions), -m 6 -b 5000.  A step = one pass of
the hot path over the whole resident batch (read packing + mapping kernels, both strand passes).
`--genome easy` is the round-1 genome (iid + four small families).

N > 1 (configs[3]): `python bench.py --gpus N` starts N ranks itself (torch.distributed.run as a child
process, before anything touches the GPU); under an external launcher it checks WORLD_SIZE == N.  Every
rank holds a full index replica and its own 50 M-read shard (weak scaling); the only collective is the
final all-reduce of the statistics block over RCCL.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     : algorithmic bytes of the implemented search per launch (128-byte line per dependent
                 gather) / HIP-event duration of the mapping kernels, vs 8 TB/s;
                 measured HBM traffic (PMC, profiles/traffic.json) beside it
  cpu_baseline : the oracle restatement (bit-exact to the reference, OpenMP on all host cores) timed on a
                 uniform sample of the batch; the same pass checks GPU records bit for bit on that sample
                 and on the hard classes (ambiguous, unmapped, deferred to the literal pass)
  extra_lines  : (N = 1) the other BASELINE configs timed in the same process on the same box:
                 configs[2] paired-end 2 x 100 bp, configs[4] 150 bp -A single-end and 2 x 150 bp PBAT
                 paired-end at -m 10 -- each with its own roofline / cpu_baseline / bit_exact_vs_gpu.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW"


def csrc_sha():
    """Content hash of the kernel sources (walt_amd/csrc, include/): profiles/traffic.json carries the hash of the
    sources its PMC counters were collected on, and roofline.traffic is dropped when the kernels have changed since
    (.git does not travel to the GPU box, so the files themselves are hashed)."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "walt_amd", "csrc")
    files = []
    for d, _, fs in os.walk(base):
        files += [os.path.join(d, f) for f in fs if f.endswith((".h", ".hip", ".cpp")) or f == "Makefile"]
    files.append(os.path.join(ROOT, "include", "walt_amd.h"))
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def traffic_entry(key):
    """profiles/traffic.json[key] when it was measured on the kernels as they are now, else None."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
    except (OSError, ValueError):
        return None
    if not t or t.get("csrc_sha") != csrc_sha():
        return None
    return t


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench] " + msg, file=sys.stderr, flush=True)


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome", choices=["hg19like", "easy"], default="hg19like",
                    help="hg19like: 93 sequences, repeat-rich (tools/synth.py); easy: the round-1 genome")
    ap.add_argument("--genome-mbp", type=float, default=None, help="synthetic genome size (default: full scale)")
    ap.add_argument("--reads", type=int, default=50_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000, help="uniform sample of the batch the oracle maps")
    ap.add_argument("--hard-sample", type=int, default=100_000, help="reads per hard class added to the exactness check")
    ap.add_argument("--max-mismatches", type=int, default=6)
    ap.add_argument("--bucket", type=int, default=5000)
    ap.add_argument("--dir-bits", type=int, default=-1)
    ap.add_argument("--contigs", type=int, default=0, help="cut the genome into this many equal contigs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["se", "pe"], default="se", help="headline leg: se = configs[1], pe = configs[2]")
    ap.add_argument("--ag", action="store_true", help="single-end -A: A-rich reads on the _GA10/_GA11 indexes")
    ap.add_argument("--pbat", action="store_true", help="paired-end -P: mate 1 is the A-rich mate (mates exchanged for mapping)")
    ap.add_argument("--top-k", type=int, default=50)
    ap.add_argument("--frag-range", type=int, default=1000)
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_lines legs (configs[2], configs[4])")
    ap.add_argument("--extra-pairs", type=int, default=0, help="pairs of the extra paired-end legs (default: --reads, 150 bp: half)")
    ap.add_argument("--extra-steps", type=int, default=10)
    ap.add_argument("--extra-contigs", type=int, default=3000,
                    help="sequences of the many-contigs extra leg (the headline genome cut into that many; 0: no such leg)")
    ap.add_argument("--ref-sample", type=int, default=2_000_000,
                    help="reads the real reference binary (oracle/_ref/walt) maps beside the oracle port at N=1 "
                         "(0 = skip; needs ~35 GB of RAM-backed scratch for the index copy)")
    ap.add_argument("--e2e-reads", type=int, default=20_000_000,
                    help="reads of the end-to-end leg (walt_amd/bin/walt FASTQ -> SAM on RAM-backed scratch, N = 1; 0 = skip)")
    ap.add_argument("--e2e-batch", type=int, default=10_000_000, help="-N of the end-to-end leg")
    ap.add_argument("--no-calibration", action="store_true",
                    help="skip cpu_baseline.calibration (configs[0]: reference binary vs oracle port at -t 1 / -t N on a chr2-length genome)")
    ap.add_argument("--calibrate", action="store_true", help="run ONLY the configs[0] calibration leg and print its JSON")
    ap.add_argument("--slot-table", action="store_true",
                    help="build the opt-in direct-mapped slot table (WALT_AMD_TABLE=1: +51.5 GB per strand at hg19 scale)")
    ap.add_argument("--seed-offset", type=int, default=0,
                    help="added to the rank in the read seeds (tests: rank r of an N-rank run == a 1-rank run at offset r)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="an option of the mapping library on every index the run opens (walt_index_set_option; A/B runs)")
    ap.add_argument("--pattern", type=int, choices=[3, 5, 7], default=3,
                    help="seed pattern (the reference's -D SEEDPATTERN3/5/7); 5 and 7 use libwalt_amd_sp5/_sp7.so")
    return ap.parse_args(argv)


def apply_opts(idx, args):
    """--opt NAME=VALUE: options of the mapping library (schedule only, never results), for A/B runs"""
    for kv in args.opt:
        name, _, value = kv.partition("=")
        idx.set_option(name, int(value))


def launch_ranks(args, argv):
    """--gpus N without a launcher: start the N ranks as a CHILD process (this process never initialises the
    GPU), forward their output and exit with their status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("[bench] starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


class Ctx:
    pass


# ------------------------------------------------------------------------------------------------ single-end
def se_leg(cx, idx, d_bases, d_off, n, read_len, max_mm, b, ag, steps, warmup, timed_barrier=True):
    """K timed steps of the single-end hot path on a resident batch; returns timing + device result tensors."""
    import numpy as np
    torch, walt_amd = cx.torch, cx.walt_amd
    dev = cx.dev
    d_out = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    d_stats = torch.zeros(4, dtype=torch.int64, device=dev)
    d_ws = torch.empty(walt_amd.lib().walt_se_workspace_bytes(n, read_len), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    idx.profile_enable(True)

    def step():
        idx.map_se_batch_device(d_bases.data_ptr(), d_off.data_ptr(), n, read_len, d_out.data_ptr(),
                                d_stats.data_ptr(), d_ws.data_ptr(), d_ws.numel(), stream=stream, ag_wildcard=ag,
                                max_mismatches=max_mm, b=b)

    for _ in range(warmup):
        step()
    cx.barrier() if timed_barrier else torch.cuda.synchronize()
    pack_ms, map_ms, detail = [], [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        p_ms, m_ms = idx.profile_last()  # waits for this step's events (same stream)
        pack_ms.append(p_ms)
        map_ms.append(m_ms)
        detail.append(idx.profile_detail())  # the same call's mapping time by kernel group (HIP events between the groups)
    cx.barrier() if timed_barrier else torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    idx.check_batch(d_ws.data_ptr(), stream)  # no invalid read went unnoticed
    st = d_stats.cpu().numpy() // (warmup + steps)  # walt_batch_stats of ONE step
    ctl = d_ws[:64 * 4].view(torch.int32).cpu().numpy()
    n_def = int(ctl[32])
    n_heavy = int(ctl[56])  # reads pass 1 handed to the heavy pass (map_se.hip: heavy_count = control word 24)
    # deferred list of the last step (workspace layout of map_se.hip: control words, statistic shards, list)
    deferred = None
    if n_def and cx.pattern == 3:
        off = 64 * 4 + 256 * 16 * 8
        deferred = (d_ws[off:off + 4 * n_def].view(torch.int32) & 0x0FFFFFFF).long()
    # the staged heavy pass's control words (map_se.hip: 8 per (chunk, round), behind the dense 2-bit reads): per chunk
    # and round [reads blocked -> next round, dense items, gather items, giants]
    stride = (n + 63) // 64 * 64
    h_off = (64 * 4 + 256 * 16 * 8 + 3 * stride * 4 + ((n * read_len) // 16 + 10) * 4 + 15) // 16 * 16
    kpat = cx.pattern
    hctl = d_ws[h_off:h_off + 4 * 64 * kpat].view(torch.int32).cpu().numpy().reshape(8, kpat, 8)
    rounds = [[[int(hctl[c, r, 4]), int(hctl[c, r, 0]), int(hctl[c, r, 1]), int(hctl[c, r, 5])] for r in range(kpat)]
              for c in range(8) if hctl[c].any()]
    return {"elapsed": elapsed, "pack_ms": pack_ms, "map_ms": map_ms, "d_out": d_out, "stats": st, "deferred": deferred,
            "n_deferred": n_def, "n_heavy": n_heavy, "d_ws": d_ws, "detail_ms": np.median(np.array(detail), axis=0).tolist(),
            "rounds": rounds}


def se_sample(cx, leg, n, n_uniform, n_hard):
    """Read indices the oracle checks: a uniform stride sample of the whole batch, plus up to n_hard reads of each
    hard class as the GPU saw them (ambiguous, unmapped, deferred to the literal pass)."""
    torch = cx.torch
    times = leg["d_out"].view(torch.int32).view(n, 4)[:, 1]
    nu = min(n_uniform, n)
    uni = (torch.arange(nu, device=cx.dev, dtype=torch.int64) * n) // nu
    hard = []
    for cls in (times >= 2, times == 0):
        ix = torch.nonzero(cls).flatten()
        if ix.numel() > n_hard:
            ix = ix[(torch.arange(n_hard, device=cx.dev, dtype=torch.int64) * ix.numel()) // n_hard]
        hard.append(ix)
    if leg["deferred"] is not None:
        ix = leg["deferred"]
        if ix.numel() > n_hard:
            ix = ix[(torch.arange(n_hard, device=cx.dev, dtype=torch.int64) * ix.numel()) // n_hard]
        hard.append(ix)
    hard = torch.unique(torch.cat(hard)) if hard else torch.zeros(0, dtype=torch.int64, device=cx.dev)
    return uni, hard


def oracle_se_jobs(cx, idx, jobs, lens, strands):
    """One export of each strand to the host serves every job (the reference also holds one strand index at a
    time, mapping.cpp:491-492).  job: dict(bases [m*L] uint8 numpy, m, read_len, max_mm, b, ag, timed_first) ->
    adds out (orc_best[m]), trace, work, cpu_s (time of the first `timed_first` reads only)."""
    import numpy as np
    import refio
    cores = cx.walt_amd.effective_cpus()
    orc = refio.oracle()
    start = np.zeros(len(lens) + 1, dtype=np.uint32)
    start[1:] = np.cumsum(lens, dtype=np.uint64).astype(np.uint32)
    for j in jobs:
        m = j["m"]
        j["out"] = np.zeros(m, dtype=refio.best_dtype)
        orc.orc_se_init(j["out"].ctypes.data, m, j["max_mm"])
        j["trace"] = np.zeros(m, dtype=refio.trace_dtype)
        j["work"] = np.zeros(1, dtype=refio.work_dtype)
        j["work_timed"] = np.zeros(1, dtype=refio.work_dtype)
        j["cpu_s"] = 0.0
        j["cores"] = cores
    for k, ch in ((0, b"+"), (1, b"-")):
        g, cnt, ix = idx.export_strand(strands[k])
        x = refio.make_orc_strand(g, cnt, ix, start)
        for j in jobs:
            L, m, mt = j["read_len"], j["m"], min(j["timed_first"], j["m"])
            offs = np.arange(m + 1, dtype=np.uint64) * L
            for lo, hi, timed in ((0, mt, True), (mt, m, False)):
                if hi <= lo:
                    continue
                w = j["work_timed"] if timed else j["work"]
                t0 = time.perf_counter()
                orc.orc_se_map_strand_trace(ctypes.addressof(x), ch, j["bases"].ctypes.data + lo * L,
                                            (offs[:hi - lo + 1]).ctypes.data, hi - lo, int(j["ag"]), j["b"], cores,
                                            j["out"].ctypes.data + lo * 16, w.ctypes.data,
                                            j["trace"].ctypes.data + lo * 32)
                if timed:
                    j["cpu_s"] += time.perf_counter() - t0
        del g, cnt, ix, x


def se_report(cx, args, leg, job, n, read_len, max_mm, b, sel_all, n_uniform, traffic_key, kernel_name):
    """cpu_baseline + roofline objects of a single-end leg from its oracle job."""
    import numpy as np
    torch, walt_amd = cx.torch, cx.walt_amd
    got = leg["d_out"].view(torch.int32).view(n, 4)[sel_all].cpu().numpy().view(walt_amd.best_match_dtype).reshape(-1)
    want = job["out"]
    same = all(np.array_equal(got[f], want[f]) for f in ("genome_pos", "times", "strand", "mismatch"))
    nu = min(n_uniform, job["m"])
    tr = job["trace"]
    wt = job["work_timed"][0]
    P, S, C = float(wt["probes"]) / nu, float(wt["steps"]) / nu, float(wt["cands"]) / nu
    tu = tr[:nu]
    big = tu["max_region"] > 4
    classes = {"sampled_uniform": int(nu), "sampled_hard": int(job["m"] - nu),
               "uniform_with_region_gt4": int(big.sum()), "uniform_with_region_gt64": int((tu["max_region"] > 64).sum()),
               "uniform_with_region_gt1000": int((tu["max_region"] > 1000).sum()),
               "uniform_over_b": int((tu["over_b"] > 0).sum()),
               "hard_with_region_gt4": int((tr[nu:]["max_region"] > 4).sum()), "hard_over_b": int((tr[nu:]["over_b"] > 0).sum()),
               "deferred_to_literal_pass_in_batch": int(leg["n_deferred"])}
    uniq = want["times"][:nu] == 1
    reg_of_unique = tu["max_region"][uniq]
    cpu = {"value": nu / job["cpu_s"], "unit": "reads/s", "cores": job["cores"], "kind": "port",
           "sample": "%d reads sampled uniformly (stride) over rank 0's batch, both strand passes, oracle restatement "
                     "with OpenMP; one strand index in host memory at a time like the reference" % nu,
           "bit_exact_vs_gpu": bool(same),
           "exactness_sample": classes,
           "unique_reads_by_largest_region": {">1": float((reg_of_unique > 1).mean()) if reg_of_unique.size else 0.0,
                                              ">10": float((reg_of_unique > 10).mean()) if reg_of_unique.size else 0.0,
                                              ">100": float((reg_of_unique > 100).mean()) if reg_of_unique.size else 0.0,
                                              ">1000": float((reg_of_unique > 1000).mean()) if reg_of_unique.size else 0.0,
                                              ">5000": float((reg_of_unique > 5000).mean()) if reg_of_unique.size else 0.0}}
    # Algorithmic bytes per read of the IMPLEMENTED search.  P probes and C verified
    # candidates are the reference algorithm's own counts (oracle).  Per probe the directory/key
    # search must read one directory pair and one run of entries: two dependent random gathers, which HBM
    # serves in whole 128-byte lines.  Per candidate it must read the genome window: for the C_big candidates
    # of regions of more than 16 slots the index holds the window once more in slot order (dense candidate
    # windows), so they stream at 32 bytes per candidate (48 above 110 bases); the other
    # C - C_big cost a 12-byte entry and one scattered 128-byte line each.  Plus the packed read (L/4 bytes) and
    # the 16-byte result.  SURVEY 8(d)'s formula prices the REFERENCE algorithm's binary-search steps and is
    # kept as `survey_8d`.
    table = bool(os.environ.get("WALT_AMD_TABLE", "0") not in ("", "0"))
    lead = args.pattern - 1  # bases a dense record holds in front of the entry's position (core.h kWinLead)
    dense_on = os.environ.get("WALT_AMD_WIN", "1") not in ("0",) and read_len <= 176 - lead
    C_big = float(tu["cands_big"].sum()) / nu if dense_on else 0.0
    rec_bytes = 32.0 if read_len <= 112 - lead else 48.0
    lines = (1.0 if table else 2.0) * P + (C - C_big)
    bytes_per_read = 128.0 * lines + 12.0 * (C - C_big) + rec_bytes * C_big + read_len / 4.0 + 16
    useful = read_len / 4.0 + 16 + P * (8 + 12) + C * (12 + read_len / 4.0 + 8)
    # N > 1: the slowest rank's kernel time (MAX over ranks of the per-rank median), so that `frac` is what every GPU
    # of the job at least reaches; counters and sample are rank 0's (every rank draws its reads from the same model)
    kern_s = float(leg.get("kern_ms_ranks_max") or np.median(leg["map_ms"])) / 1e3
    achieved = bytes_per_read * n / kern_s
    survey = read_len + 16 + 8 * P + S * 4.25 + C * (4 + read_len / 4.0)
    traffic = None
    t = traffic_entry(traffic_key)
    if t and t.get("reads_per_launch") == n and t.get("genome") == args.genome and not args.contigs and not table \
            and args.pattern == 3 and t.get("genome_bp") == cx.genome_bp:
        traffic = t.get("hbm_bytes_per_launch")
    # the kernel groups of the call, timed by HIP events between them (walt_profile_detail); the region verifier's
    # algorithmic bytes are its dense records alone, so it has a roofline of its own
    dms = leg.get("detail_ms") or [0.0, 0.0, 0.0, 0.0]
    by_kernel = {"k_map_se pass 1": {"ms": dms[0]}, "k_map_se heavy stages": {"ms": dms[1]},
                 "k_se_verify": {"ms": dms[2]}, "k_map_se_literal (+ sort)": {"ms": dms[3]}}
    if dms[2] > 0 and C_big > 0:
        vb = rec_bytes * C_big * n
        by_kernel["k_se_verify"].update({"algorithmic_bytes": vb, "achieved": vb / (dms[2] / 1e3) / 1e9, "unit": "GB/s",
                                         "frac": vb / (dms[2] / 1e3) / HBM_PEAK})
    # `frac` / `achieved`: STRICT algorithmic bytes (what the search needs byte by byte: `useful`); `frac_lines` /
    # `achieved_lines`: the same work priced at one 128-byte line per dependent gather (what HBM has to deliver for it).
    # Scalars first: the driver's record keeps scalars only.
    strict = useful * n / kern_s
    names = ("k_map_se pass 1", "k_se_stage heavy stages", "k_se_verify", "k_map_se_literal (+ sort)")
    dom = int(np.argmax(dms)) if max(dms) > 0 else 0
    vb = rec_bytes * C_big * n                          # the verifier's bytes: its dense records, nothing else
    look_ms = dms[0] + dms[1] + dms[3]                  # every kernel that is not the verifier: dependent gathers
    look_lines = lines * n                              # 128-byte lines those kernels need (2 per probe, 1 per candidate outside the windows)
    look_bytes = bytes_per_read * n - vb
    dom_frac = None
    if dms[dom] > 0:
        dom_frac = (vb / (dms[2] / 1e3) / HBM_PEAK) if dom == 2 else (look_bytes / (look_ms / 1e3) / HBM_PEAK)
    roof = {"bound": "hbm", "achieved": strict / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": strict / HBM_PEAK, "frac_lines": achieved / HBM_PEAK, "achieved_lines": achieved / 1e9,
            "traffic": traffic, "traffic_frac": (traffic / kern_s / HBM_PEAK) if traffic else None,
            "kernel": kernel_name, "kernel_ms_median": kern_s * 1e3, "kernel_ms_min": float(np.min(leg["map_ms"])),
            "algorithmic_bytes_per_read": useful, "line_bytes_per_read": bytes_per_read,
            "dominant_kernel": names[dom], "dominant_kernel_ms": dms[dom], "dominant_kernel_frac": dom_frac,
            "dominant_kernel_frac_is": "dense-record bytes / its time" if dom == 2 else
                                       "line bytes of all look-up kernels (pass 1 + stages + literal) / their time",
            "pass1_ms": dms[0], "stage_ms": dms[1], "verify_ms": dms[2], "literal_ms": dms[3],
            "verify_frac": (vb / (dms[2] / 1e3) / HBM_PEAK) if dms[2] > 0 else None,
            "lookup_frac_lines": (look_bytes / (look_ms / 1e3) / HBM_PEAK) if look_ms > 0 else None,
            "gather_lines_per_s": (look_lines / (look_ms / 1e3)) if look_ms > 0 else None,
            "gather_ceiling_lines_per_s": None,  # filled in by the caller (tools/gather_calib in the same run)
            "probes_per_read": P, "candidates_per_read": C, "candidates_in_regions_gt16_per_read": C_big,
            "granularity": "frac: bytes (%d B/read: read + result + 20 B per probe + 45 B per candidate); frac_lines: 128-byte line "
                           "per dependent gather (%d per probe, 1 per candidate outside the dense windows) + 12-byte entry / "
                           "%d-byte dense record per candidate + streamed read/result bytes" % (
                               int(useful), 1 if table else 2, int(rec_bytes)),
            "by_kernel": by_kernel,
            "per_read": {"probes": P, "search_steps": S, "candidates": C, "candidates_in_regions_gt16": C_big},
            "per_read_source": "oracle counters on this run's uniform sample",
            "survey_8d": {"bytes_per_read": survey, "achieved": survey * n / kern_s / 1e9,
                          "frac": survey * n / kern_s / HBM_PEAK,
                          "note": "bytes of the reference algorithm (binary-search steps S); not a bound on this "
                                  "kernel, which replaces them by a directory lookup"}}
    return cpu, roof


def write_fastq(path, host_bases, n, read_len):
    """@r<9 digits>\\n<seq>\\n+\\n<qual>\\n per read, built with numpy (no per-read Python)."""
    import numpy as np
    rec = np.dtype([("at", "S2"), ("num", "S9"), ("nl0", "S1"), ("seq", "S%d" % read_len), ("mid", "S3"),
                    ("qual", "S%d" % read_len), ("nl1", "S1")])
    with open(path, "wb") as f:
        for s0 in range(0, n, 1 << 20):
            m = min(1 << 20, n - s0)
            a = np.zeros(m, dtype=rec)
            a["at"], a["nl0"], a["mid"], a["nl1"] = b"@r", b"\n", b"\n+\n", b"\n"
            a["num"] = np.char.zfill(np.arange(s0, s0 + m).astype("S9"), 9)
            a["seq"] = host_bases[s0 * read_len:(s0 + m) * read_len].view("S%d" % read_len)
            a["qual"] = b"I" * read_len
            f.write(a.tobytes())


def read_mapstats(path):
    stats = {}
    for ln in open(path):
        k, _, v = ln.strip().partition(":")
        if v.strip().lstrip("-").replace(".", "", 1).isdigit():
            stats[k.strip()] = v.strip()
    return stats


def e2e_leg(cx, scratch, dbi, host_bases, times_host, read_len, n_e2e, max_mm, b, cores, batch):
    """End to end through the product's own command line, index files and reads on RAM-backed scratch:
    walt_amd/bin/walt -i <dbindex> -r <fastq> -o <sam> -sam -a -u (FASTQ parse -> N draws -> upload -> kernels -> SAM
    formatting -> write), several -N batches.  The bench's own index has been closed by now (the GPU is the binary's).
    reads/s is quoted with the one-time index load excluded (the reference reloads its index per batch,
    mapping.cpp:491-492) and, beside it, on the whole wall clock; the stage times are the binary's own (-v)."""
    import re
    import subprocess

    import numpy as np
    our_bin = os.path.join(ROOT, "walt_amd", "bin", "walt")
    if not os.path.exists(our_bin):
        return {"skipped": "walt_amd/bin/walt is not built"}
    fq = os.path.join(scratch, "e2e.fastq")
    t0 = time.perf_counter()
    write_fastq(fq, host_bases, n_e2e, read_len)
    t_fq = time.perf_counter() - t0
    out = os.path.join(scratch, "e2e.sam")
    cmd = [our_bin, "-i", dbi, "-r", fq, "-o", out, "-m", str(max_mm), "-b", str(b), "-a", "-u", "-sam", "-t", str(cores),
           "-N", str(batch), "-v"]
    env = dict(os.environ)
    env["WALT_AMD_HOST_CEILING"] = "1"  # -v then also measures what the host side can copy and store at all (walt_main.cpp host_ceiling)
    t0 = time.perf_counter()
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
    wall = time.perf_counter() - t0
    if pr.returncode != 0:
        return {"skipped": "bin/walt failed: " + pr.stdout[-300:]}

    def parse_stages(text):
        st_ = {}
        for ln in text.splitlines():
            if ln.startswith("[walt_amd:"):
                for key, pat in (("index_s", r"index ([0-9.]+) s"), ("ingest_not_hidden_s", r"previous batch ([0-9.]+) s"),
                                 ("map_s", r"map ([0-9.]+) s"), ("format_s", r"format ([0-9.]+) s"), ("write_s", r"write ([0-9.]+) s"),
                                 ("since_main_s", r"since main ([0-9.]+) s")):
                    mt = re.search(pat, ln)
                    if mt:
                        st_[key] = float(mt.group(1))
        return st_
    stages = parse_stages(pr.stdout)
    ceiling = {}
    mt = re.search(r"host ceiling: memcpy ([0-9.]+) GB/s on (\d+) threads, pwrite into one file ([0-9.]+) GB/s", pr.stdout)
    if mt:
        ceiling = {"memcpy_gb_per_s": float(mt.group(1)), "threads": int(mt.group(2)), "pwrite_one_file_gb_per_s": float(mt.group(3))}
        wall -= 0.0  # (the measurement runs after the last output file is closed; its time is inside `wall`, not inside the stages)
    st = read_mapstats(out + ".mapstats")
    t = times_host[:n_e2e]
    mine = {"unique": int((t == 1).sum()), "ambiguous": int((t >= 2).sum()), "unmapped": int((t == 0).sum())}
    key_of = {"unique": "unique_mapped_reads", "ambiguous": "ambiguous_mapped_reads", "unmapped": "unmapped_reads"}
    same = all(int(st.get(key_of[k], st.get(k, -1))) == v for k, v in mine.items())
    sam_bytes = os.path.getsize(out)
    # the binary's own clock ("since main", printed before the ceiling measurement) when it is there: the wall clock of
    # this process also holds the ceiling measurement and the child's start
    run_s = stages.get("since_main_s", wall)
    resident = run_s - stages.get("index_s", 0.0)
    # several index replicas (-g 0,0: two on this box's one GPU when they fit): the multi-device host path, timed
    multi = None
    try:
        out2 = os.path.join(scratch, "e2e_g00.sam")
        n2 = min(n_e2e, batch)
        fq2 = os.path.join(scratch, "e2e_g00.fastq")
        write_fastq(fq2, host_bases, n2, read_len)
        cmd2 = [our_bin, "-i", dbi, "-r", fq2, "-o", out2, "-m", str(max_mm), "-b", str(b), "-a", "-u", "-sam", "-t", str(cores),
                "-N", str(batch), "-v", "-g", "0,0"]
        t0 = time.perf_counter()
        pr2 = subprocess.run(cmd2, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        w2 = time.perf_counter() - t0
        if pr2.returncode == 0:
            st2 = parse_stages(pr2.stdout)
            r2 = st2.get("since_main_s", w2) - st2.get("index_s", 0.0)
            multi = {"devices": "0,0", "reads": n2, "wall_s": w2, "stages_s": st2, "reads_per_s_index_load_excluded": n2 / max(1e-9, r2)}
        else:
            multi = {"skipped": "bin/walt -g 0,0 failed (two index replicas on one GPU): " + pr2.stdout[-200:]}
        for f in (fq2, out2, out2 + ".mapstats"):
            try:
                os.remove(f)
            except OSError:
                pass
    except Exception as e:
        multi = {"skipped": "%s: %s" % (type(e).__name__, e)}
    line = {"metric": "reads/s end to end, FASTQ -> SAM through walt_amd/bin/walt (%d bp single-end, -m %d -b %d -a -u -sam), "
                      "index load excluded" % (read_len, max_mm, b),
            "value": n_e2e / resident, "unit": "reads/s", "n_gpus": 1, "higher_is_better": True, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "the first %d reads of the headline batch as a FASTQ file (%.1f GB) and the headline index as "
                                   ".dbindex files, both on RAM-backed scratch; -N %d (%d batches), -t %d host threads" % (
                                       n_e2e, os.path.getsize(fq) / 1e9, batch, (n_e2e + batch - 1) // batch, cores)},
            "wall_s": run_s, "reads_per_s_whole_wall": n_e2e / run_s, "stages_s": stages, "sam_bytes": sam_bytes,
            "fastq_write_s": t_fq, "mapstats_equal_gpu_records": bool(same), "host_threads": cores,
            # the host's own limits, measured by the same binary in the same run: the format stage against the threads'
            # memcpy rate, the write stage against pwrite into one file
            "host_ceiling": ceiling,
            "format_gb_per_s": (sam_bytes / stages["format_s"] / 1e9) if stages.get("format_s") else None,
            "write_gb_per_s": (sam_bytes / stages["write_s"] / 1e9) if stages.get("write_s") else None,
            "write_frac_of_pwrite_ceiling": (sam_bytes / stages["write_s"] / 1e9 / ceiling["pwrite_one_file_gb_per_s"])
            if stages.get("write_s") and ceiling.get("pwrite_one_file_gb_per_s") else None,
            "two_replicas_one_gpu": multi}
    for f in (fq, out):
        try:
            os.remove(f)
        except OSError:
            pass
    return line


def reference_binary_leg(cx, idx, d_bases, d_out, read_len, n_ref, max_mm, b, cores, keep=None, extra_bytes=0):
    """The REAL reference binary (oracle/_ref/walt, built from the reference sources by oracle/Makefile.ref;
    test infrastructure) on the same box in the same run: the resident index is written in the reference's
    .dbindex format to a RAM-backed scratch directory, the first n_ref reads of the batch go to a FASTQ file,
    `walt -t <cores>` maps them, and its .mapstats is compared with the GPU's records for the same reads."""
    import shutil
    import subprocess
    import tempfile

    import numpy as np
    torch = cx.torch
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "walt")
    if not os.path.exists(ref_bin):
        return {"skipped": "oracle/_ref/walt is not built"}
    need = 2 * (idx.genome_len + 4 * (idx.index_size(0) + (1 << 24) + 8)) + n_ref * (2 * read_len + 20) * 2 + extra_bytes
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
        write_fastq(fq, host, n_ref, read_len)
        out = os.path.join(scratch, "ref.mr")
        cmd = [ref_bin, "-i", dbi, "-r", fq, "-o", out, "-m", str(max_mm), "-b", str(b), "-t", str(cores)]
        t0 = time.perf_counter()
        pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        wall = time.perf_counter() - t0
        if pr.returncode != 0:
            return {"skipped": "reference binary failed: " + pr.stdout[-300:]}
        stats = read_mapstats(out + ".mapstats")
        for f in (fq, out):
            os.remove(f)
        times = d_out[:n_ref * 16].view(torch.int32).view(n_ref, 4)[:, 1]
        mine = {"unique": int((times == 1).sum().item()), "ambiguous": int((times >= 2).sum().item()),
                "unmapped": int((times == 0).sum().item())}
        key_of = {"unique": "unique_mapped_reads", "ambiguous": "ambiguous_mapped_reads", "unmapped": "unmapped_reads"}
        same = all(int(stats.get(key_of[k], stats.get(k, -1))) == v for k, v in mine.items())
        if keep is not None:
            keep["scratch"], keep["dbi"] = scratch, dbi
        return {"value": n_ref / wall, "unit": "reads/s", "cores": cores, "kind": "reference",
                "sample": "oracle/_ref/walt -t %d on the first %d reads of the batch, wall clock of the whole run "
                          "including its read of both strand index files (%.0f GB, RAM-backed) -- the reference "
                          "reloads them for every batch (mapping.cpp:491-492)" % (
                              cores, n_ref, 2 * (idx.genome_len + 4 * (idx.index_size(0) + (1 << 24) + 8)) / 1e9),
                "wall_s": wall, "index_write_s": t_write, "mapstats_equal_gpu": bool(same)}
    finally:
        if keep is None or "scratch" not in keep:
            shutil.rmtree(scratch, ignore_errors=True)


def calibration_leg(cx, local, cores, n=100_000, read_len=100, max_mm=6, b=5000):
    """BASELINE configs[0]: a chr2-length genome (243,199,373 bp, one sequence; the hg19-like
    families scaled to it), 100,000 x 100 bp single-end C->T reads, -m 6 -- the REAL reference binary (oracle/_ref/walt,
    test infrastructure) at -t 1 and -t <cores> beside the oracle port at the same thread counts, on this box in this run.
    What it pins: (i) the three SAM texts are the same bytes -- reference -t 1, reference -t N, and the port's records
    through the restated writer (tests/refio.py) -- and the GPU's records equal the port's; (ii) the port is not slower
    than real WALT, as a number: reads/s of the mapping work alone, which for the binary is its wall clock minus the wall
    clock of the same command on ONE read (index read, start-up and allocation: the reference reloads its index per
    batch, mapping.cpp:491-492), for the port the two strand passes over indexes already in memory."""
    import shutil
    import subprocess
    import tempfile

    import numpy as np
    import refio
    import synth
    torch, walt_amd = cx.torch, cx.walt_amd
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "walt")
    if not os.path.exists(ref_bin):
        return {"skipped": "oracle/_ref/walt is not built"}
    dev = cx.dev
    genome_ascii, lens, names = synth.make_genome(torch, dev, 1.0, seed=2, kind="chr2like")
    idx = walt_amd.Index.build_device(genome_ascii.data_ptr(), lens, names, device=local, strands=walt_amd.STRANDS_CT)
    d_bases, d_off = synth.make_reads(torch, dev, genome_ascii, n, read_len, seed=7000, lowq=False)
    del genome_ascii
    torch.cuda.synchronize()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    scratch = tempfile.mkdtemp(prefix="walt_amd_calib_", dir=base)
    try:
        # the GPU's records for the same reads, through the C ABI
        d_out = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
        d_stats = torch.zeros(4, dtype=torch.int64, device=dev)
        d_ws = torch.empty(walt_amd.lib().walt_se_workspace_bytes(n, read_len), dtype=torch.uint8, device=dev)
        idx.map_se_batch_device(d_bases.data_ptr(), d_off.data_ptr(), n, read_len, d_out.data_ptr(), d_stats.data_ptr(),
                                d_ws.data_ptr(), d_ws.numel(), stream=torch.cuda.current_stream().cuda_stream, ag_wildcard=False,
                                max_mismatches=max_mm, b=b)
        torch.cuda.synchronize()
        gpu = d_out.cpu().numpy().view(walt_amd.best_match_dtype).reshape(-1)
        host = d_bases.cpu().numpy()
        dbi = os.path.join(scratch, "chr2.dbindex")
        idx.write(dbi)
        for sfx in ("_GA10", "_GA11"):  # the binary only checks that all four strand files exist
            open(dbi + sfx, "wb").close()
        fq, fq1 = os.path.join(scratch, "reads.fastq"), os.path.join(scratch, "one.fastq")
        write_fastq(fq, host, n, read_len)
        write_fastq(fq1, host, 1, read_len)
        ref = {}
        sams = {}
        for t in sorted({1, cores}):
            walls = []
            for f, tag in ((fq, "full"), (fq1, "one")):
                o = os.path.join(scratch, "ref_%s_t%d.sam" % (tag, t))
                cmd = [ref_bin, "-i", dbi, "-r", f, "-o", o, "-m", str(max_mm), "-b", str(b), "-a", "-u", "-sam", "-t", str(t)]
                t0 = time.perf_counter()
                pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                walls.append(time.perf_counter() - t0)
                if pr.returncode != 0:
                    return {"skipped": "reference binary failed: " + pr.stdout[-300:]}
            sams[t] = open(os.path.join(scratch, "ref_full_t%d.sam" % t), "rb").read()
            ref[t] = {"wall_s": walls[0], "wall_one_read_s": walls[1],
                      "reads_per_s": n / max(1e-9, walls[0] - walls[1])}
        # the port: both strand passes with the index in memory, at the same thread counts
        orc = refio.oracle()
        start = np.zeros(len(lens) + 1, dtype=np.uint32)
        start[1:] = np.cumsum(lens, dtype=np.uint64).astype(np.uint32)
        strands = [idx.export_strand(k) for k in (0, 1)]
        xs = [refio.make_orc_strand(g, cnt, ix, start) for g, cnt, ix in strands]
        offs = np.arange(n + 1, dtype=np.uint64) * read_len
        port = {}
        rec = None
        for t in sorted({1, cores}):
            rec = np.zeros(n, dtype=refio.best_dtype)
            orc.orc_se_init(rec.ctypes.data, n, max_mm)
            work = np.zeros(1, dtype=refio.work_dtype)
            t0 = time.perf_counter()
            for k, ch in ((0, b"+"), (1, b"-")):
                orc.orc_se_map_strand_trace(ctypes.addressof(xs[k]), ch, host.ctypes.data, offs.ctypes.data, n, 0, b, t,
                                            rec.ctypes.data, work.ctypes.data, None)  # no per-read trace: the plain port
            dt = time.perf_counter() - t0
            port[t] = {"map_s": dt, "reads_per_s": n / dt}
        same_gpu = all(np.array_equal(gpu[f], rec[f]) for f in ("genome_pos", "times", "strand", "mismatch"))

        class Db:  # what refio's writers read
            pass
        db = Db()
        db.names, db.lengths, db.start_index, db.n_chrom = list(names), np.array(lens, dtype=np.uint32), start, len(lens)
        db.chrom_of = lambda pos: int(np.searchsorted(start, pos, side="right") - 1) if len(lens) > 1 else 0
        seqs = host.view("S%d" % read_len)
        qual = "I" * read_len
        out = [refio.sam_header(db)]
        for i in range(n):
            out.append(refio.se_sam_line(db, rec[i], "r%09d" % i, seqs[i].decode(), qual, True, True))
        port_sam = "".join(out).encode()
        t_lo, t_hi = min(ref), max(ref)
        return {"workload": "configs[0]: chr2-length synthetic genome (%d bp, 1 sequence), %d x %d bp single-end C->T reads, "
                            "-m %d -b %d -a -u -sam" % (sum(lens), n, read_len, max_mm, b),
                "reference_binary": {"t%d" % t: ref[t] for t in ref}, "port": {"t%d" % t: port[t] for t in port},
                "port_over_reference": {"t%d" % t: port[t]["reads_per_s"] / ref[t]["reads_per_s"] for t in ref},
                "sam_reference_t%d_equals_t%d" % (t_lo, t_hi): bool(sams[t_lo] == sams[t_hi]),
                "sam_port_equals_reference": bool(port_sam == sams[t_lo]), "sam_bytes": len(sams[t_lo]),
                "gpu_records_equal_port": bool(same_gpu),
                "unique_frac": float((rec["times"] == 1).mean()),
                "note": "reads/s of the mapping work alone: the binary's wall clock minus the same command on one read "
                        "(index read, start-up); the port maps with both strand indexes already in memory"}
    finally:
        idx.close()
        shutil.rmtree(scratch, ignore_errors=True)


def gather_ceiling(table_gb=64.0):
    """The device's random-line ceiling, measured in this run: tools/gather_calib (a diagnostic micro-benchmark, built by
    __graft_entry__.build) gathers 12-byte elements at random over a table of `table_gb` GB, one independent gather after
    the other per lane -- every gather is one 128-byte line from HBM.  Run as a child process while the bench holds no
    index (the table needs the memory).  Returns lines per second or a dict with the reason it could not run."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "gather_calib")
    if not os.path.exists(exe):
        return {"skipped": "tools/gather_calib is not built"}
    try:
        pr = subprocess.run([exe, str(table_gb), "3", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    except Exception as e:
        return {"skipped": "%s: %s" % (type(e).__name__, e)}
    mt = re.search(r"([0-9.]+) G gathers/s", pr.stdout)
    if pr.returncode != 0 or not mt:
        return {"skipped": "gather_calib: " + pr.stdout[-200:]}
    return float(mt.group(1)) * 1e9


# ------------------------------------------------------------------------------------------------ paired-end
def pe_leg(cx, idx, d1, d2, d_off, n, read_len, max_mm, b, top_k, frag_range, steps, warmup, timed_barrier=True):
    torch, walt_amd = cx.torch, cx.walt_amd
    dev = cx.dev
    torch.cuda.empty_cache()  # the read generator's cached blocks back to the device: the library sizes its passes by what is free
    d_out = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    d_stats = torch.zeros(8, dtype=torch.int64, device=dev)
    # what the call uses best on this device now: the larger passes when it has the room (walt_pe_workspace_bytes_best)
    d_ws = torch.empty(idx.pe_workspace_bytes(n, read_len, top_k), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        idx.map_pe_batch_device(d1.data_ptr(), d_off.data_ptr(), d2.data_ptr(), d_off.data_ptr(), n, read_len,
                                d_out.data_ptr(), d_stats.data_ptr(), d_ws.data_ptr(), d_ws.numel(), stream=stream,
                                max_mismatches=max_mm, b=b, top_k=top_k, frag_range=frag_range)

    for _ in range(warmup):
        step()
    cx.barrier() if timed_barrier else torch.cuda.synchronize()
    per_step = []
    t0 = time.perf_counter()
    for _ in range(steps):
        ts = time.perf_counter()
        step()
        torch.cuda.synchronize()
        per_step.append(time.perf_counter() - ts)
    cx.barrier() if timed_barrier else torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    idx.check_batch(d_ws.data_ptr(), stream)
    st = d_stats.cpu().numpy() // (warmup + steps)
    ctl = d_ws[:192 * 4].view(torch.int32).cpu().numpy()  # control words of the last pass (map_pe.hip carve_pe)
    return {"elapsed": elapsed, "per_step": per_step, "d_out": d_out, "stats": st,
            "lists": {"literal": [int(ctl[64]), int(ctl[96])], "literal_round_in_one_launch": [int(ctl[67]), int(ctl[99])],
                      "overflowed_small_heaps": [int(ctl[89]), int(ctl[121])],
                      "staged": [int(ctl[88]), int(ctl[120])], "staged_fallback": [int(ctl[90]), int(ctl[122])],
                      "staged_items": [[int(ctl[92]), int(ctl[93])], [int(ctl[124]), int(ctl[125])]],
                      "heavy_pairs": int(ctl[128])}}


def oracle_pe_jobs(cx, idx, jobs, lens):
    """job: dict(m1, m2 numpy bases, m, nu, read_len, max_mm, b, top_k, frag_range) -> out (pair records), cpu_s, works.
    The first `nu` pairs of a job are its UNIFORM sample: only they are timed (cpu_s) and counted (works: probes,
    candidates, candidates in regions of more than 16 slots) -- the hard classes behind them are mapped for the
    exactness check alone (they are the candidate-rich pairs; timing or counting them would bias both figures).
    The two strand indexes of one mate are in host memory at a time; both exports serve every job."""
    import numpy as np
    import refio
    cores = cx.walt_amd.effective_cpus()
    orc = refio.oracle()
    start = np.zeros(len(lens) + 1, dtype=np.uint32)
    start[1:] = np.cumsum(lens, dtype=np.uint64).astype(np.uint32)
    for j in jobs:
        j["ranked"], j["counts"], j["works"], j["cpu_s"], j["cores"] = [], [], [], 0.0, cores
    for mate in (0, 1):
        keep = []
        arr = (refio.OrcStrand * 2)()
        for k in range(2):
            g, cnt, ix = idx.export_strand(2 * mate + k)
            keep.append((g, cnt, ix))
            x = refio.make_orc_strand(g, cnt, ix, start)
            for f, _ in refio.OrcStrand._fields_:
                setattr(arr[k], f, getattr(x, f))
        for j in jobs:
            m, L, top_k, nu = j["m"], j["read_len"], j["top_k"], min(j["nu"], j["m"])
            offs = np.arange(m + 1, dtype=np.uint64) * L
            r = np.zeros((m, top_k), dtype=refio.cand_dtype)
            c = np.zeros(m, dtype=np.uint32)
            bases = np.ascontiguousarray(j["m1"] if mate == 0 else j["m2"])
            for lo, hi, timed in ((0, nu, True), (nu, m, False)):
                if hi <= lo:
                    continue
                work = np.zeros(1, dtype=refio.work_dtype)
                t0 = time.perf_counter()
                orc.orc_pe_topk_batch(ctypes.addressof(arr), bases.ctypes.data + lo * L, offs[:hi - lo + 1].ctypes.data,
                                      hi - lo, mate, j["max_mm"], j["b"], top_k, cores,
                                      r.ctypes.data + lo * top_k * r.dtype.itemsize, c.ctypes.data + lo * 4, work.ctypes.data)
                if timed:
                    j["cpu_s"] += time.perf_counter() - t0
                    j["works"].append((float(work[0]["probes"]), float(work[0]["cands"]), float(work[0]["cands_big"])))
            j["ranked"].append(r)
            j["counts"].append(c)
        del keep, arr
    for j in jobs:
        m, L, nu, top_k = j["m"], j["read_len"], min(j["nu"], j["m"]), j["top_k"]
        offs = np.arange(m + 1, dtype=np.uint64) * L
        out = np.zeros(m, dtype=refio.pair_dtype)
        r0, r1, c0, c1 = j["ranked"][0], j["ranked"][1], j["counts"][0], j["counts"][1]
        for lo, hi, timed in ((0, nu, True), (nu, m, False)):
            if hi <= lo:
                continue
            t0 = time.perf_counter()
            orc.orc_pe_merge_batch(r0.ctypes.data + lo * top_k * r0.dtype.itemsize, c0.ctypes.data + lo * 4,
                                   r1.ctypes.data + lo * top_k * r1.dtype.itemsize, c1.ctypes.data + lo * 4, top_k,
                                   offs[:hi - lo + 1].ctypes.data, offs[:hi - lo + 1].ctypes.data, hi - lo,
                                   start.ctypes.data, len(lens), j["frag_range"], j["max_mm"],
                                   out.ctypes.data + lo * out.dtype.itemsize)
            if timed:
                j["cpu_s"] += time.perf_counter() - t0
        j["out"] = out


def pe_sample(cx, leg, n, n_uniform, n_hard):
    torch = cx.torch
    bt = leg["d_out"].view(torch.int32).view(n, 16)[:, 8]
    nu = min(n_uniform, n)
    uni = (torch.arange(nu, device=cx.dev, dtype=torch.int64) * n) // nu
    hard = []
    for cls in (bt >= 2, bt == 0):
        ix = torch.nonzero(cls).flatten()
        if ix.numel() > n_hard:
            ix = ix[(torch.arange(n_hard, device=cx.dev, dtype=torch.int64) * ix.numel()) // n_hard]
        hard.append(ix)
    return uni, torch.unique(torch.cat(hard))


def pe_report(cx, args, leg, job, n, read_len, sel_all, nu, traffic_key):
    import numpy as np
    torch, walt_amd = cx.torch, cx.walt_amd
    got = leg["d_out"].view(torch.int32).view(n, 16)[sel_all].cpu().numpy().view(walt_amd.pair_result_dtype).reshape(-1)
    want = job["out"]
    same = all(np.array_equal(got[f], want[f]) for f in ("best_times", "frag_len", "best_i", "best_j", "pair_mm"))
    for m in ("m1", "m2"):
        same = same and all(np.array_equal(got[m][f], want[m][f]) for f in ("genome_pos", "times", "strand", "mismatch"))
    m_all = job["m"]
    nu = min(nu, m_all)
    cpu = {"value": nu / job["cpu_s"], "unit": "pairs/s", "cores": job["cores"], "kind": "port",
           "sample": "%d pairs sampled uniformly (stride) over rank 0's batch: oracle restatement of PairEndMapping on both "
                     "mates and strands + pair merge, OpenMP, two strand indexes in host memory at a time.  %d more pairs from "
                     "the hard classes (ambiguous / unpaired) are mapped for the exactness check only -- neither timed nor "
                     "counted" % (nu, m_all - nu),
           "bit_exact_vs_gpu": bool(same)}
    # Same accounting as the single-end line: per pair, P probes and C candidates over both mates (oracle
    # counters on the UNIFORM sample) -> 2 P dependent gathers of one 128-byte line each; a candidate of a region of
    # more than 16 slots streams as a dense record (32 bytes, 48 above 110 bases), any other costs a 12-byte entry and
    # a scattered 128-byte line; plus per mate the ranked list written by the top-k kernel and read back by the merge
    # (one line each way), the packed reads and the 64-byte pair record.  Time = the whole step (the mates' kernels
    # overlap on several streams).
    P = sum(w[0] for w in job["works"]) / nu
    C = sum(w[1] for w in job["works"]) / nu
    C_big = sum(w[2] for w in job["works"]) / nu if (os.environ.get("WALT_AMD_WIN", "1") != "0" and read_len <= 174) else 0.0
    rec_bytes = 32.0 if read_len <= 110 else 48.0
    bytes_per_pair = 128.0 * (2.0 * P + (C - C_big) + 4.0) + 12.0 * (C - C_big) + rec_bytes * C_big + 2 * read_len / 4.0 + 64
    step_s = float(leg.get("step_s_ranks_max") or np.median(leg["per_step"]))  # N > 1: the slowest rank's
    traffic = None
    t = traffic_entry(traffic_key)
    if t and t.get("pairs_per_step") == n and t.get("genome") == args.genome and t.get("genome_bp") == cx.genome_bp \
            and args.pattern == 3 and not args.contigs:
        traffic = t.get("hbm_bytes_per_step")
    # strict bytes (as the single-end line's `frac`): reads + pair record + 20 B per probe + (12-byte entry + window + 8) per
    # candidate + the ranked lists written and read back (12 bytes per kept candidate and mate, at most top_k)
    top_k = job["top_k"]
    useful = 2 * read_len / 4.0 + 64 + P * (8 + 12) + C * (12 + read_len / 4.0 + 8) + 2 * 2 * 12.0 * min(C / 2.0, float(top_k))
    roof = {"bound": "hbm", "achieved": useful * n / step_s / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": useful * n / step_s / HBM_PEAK, "frac_lines": bytes_per_pair * n / step_s / HBM_PEAK,
            "achieved_lines": bytes_per_pair * n / step_s / 1e9, "traffic": traffic,
            "traffic_frac": (traffic / step_s / HBM_PEAK) if traffic else None,
            "kernel": "whole paired-end step (k_pe_topk_dual, staged k_pe_stage / k_pe_verify / k_pe_push of both mates, k_pe_merge)",
            "step_ms_median": step_s * 1e3, "step_ms_min": float(np.min(leg["per_step"])) * 1e3,
            "algorithmic_bytes_per_pair": useful, "useful_bytes_per_pair": useful, "line_bytes_per_pair": bytes_per_pair,
            "probes_per_pair": P, "candidates_per_pair": C, "candidates_in_regions_gt16_per_pair": C_big,
            "granularity": "frac: bytes; frac_lines: 128-byte line per dependent gather (2 per probe, 1 per candidate outside the "
                           "dense windows, 2 per mate for the ranked list) + 12-byte entry / %d-byte dense record per candidate + "
                           "streamed reads / pair record" % int(rec_bytes),
            "per_pair": {"probes": P, "candidates": C, "candidates_in_regions_gt16": C_big}}
    return cpu, roof


# ------------------------------------------------------------------------------------------------ main
def dryrun(args):
    """WALT_AMD_BENCH_DRYRUN=1 (CPU test of the launch path): rendezvous over gloo, world-size check, one
    all-reduce of a statistics vector; no GPU work, not a measurement."""
    import torch
    import torch.distributed as dist
    from walt_amd import dist as wdist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world))
    if world > 1:
        dist.init_process_group("gloo")
    vec = wdist.allreduce_stats(torch.tensor([1, rank, 10 * (rank + 1)], dtype=torch.int64))
    if rank == 0:
        print(json.dumps({"dryrun": True, "n_gpus": world, "stats": vec.tolist()}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def worker(args):
    # the paired-end path keeps several streams busy (two mates x two pipeline slots); the HIP runtime
    # multiplexes streams onto 4 hardware queues unless told otherwise, before it initialises
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if args.slot_table:
        os.environ["WALT_AMD_TABLE"] = "1"
    import numpy as np
    import walt_amd  # loads the HIP library (and the HIP runtime torch will share)
    import synth
    walt_amd.set_pattern(args.pattern)
    walt_amd.lib()
    import torch
    from walt_amd import dist as wdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world))
    dist = None
    shared_gpu = bool(os.environ.get("WALT_AMD_BENCH_SHARE_GPU"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
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

    cx = Ctx()
    cx.torch, cx.walt_amd, cx.dev, cx.pattern = torch, walt_amd, dev, args.pattern
    if args.calibrate:
        import refio
        refio.set_pattern(args.pattern)
        print(json.dumps({"calibration": calibration_leg(cx, local, walt_amd.effective_cpus())}), flush=True)
        return

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    cx.barrier = barrier
    # the oracle legs (cpu_baseline, exactness, the counters of the roofline) run on rank 0, on its own shard, at every
    # N: a line without them is unmeasured.  One strand export at a time is ~15 GB of host memory.  The other ranks
    # wait at the final barrier.
    run_cpu = rank == 0 and not args.no_cpu_baseline
    if run_cpu:
        import refio
        refio.set_pattern(args.pattern)

    full = synth.HG19_TOTAL if args.genome == "hg19like" else sum(synth.HG19_CHROMS)
    scale = 1.0 if args.genome_mbp is None else args.genome_mbp * 1e6 / full
    t0 = time.perf_counter()
    genome_ascii, lens, names = synth.make_genome(torch, dev, scale, seed=2, kind=args.genome, contigs=args.contigs)
    torch.cuda.synchronize()
    cx.genome_bp = int(sum(lens))
    log("genome (%s): %d bp in %d sequences (%.1f s)" % (args.genome, sum(lens), len(lens), time.perf_counter() - t0))
    torch.cuda.empty_cache()  # hand the generator's cached blocks back: the library allocates with hipMalloc
    genome_desc = ("hg19-like synthetic genome (%d bp, %d sequences; Alu-/L1-like families, segmental duplications, "
                   "satellites, simple repeats)" if args.genome == "hg19like" else
                   "round-1 synthetic genome (%d bp, %d chromosomes, iid + four small repeat families)") % (sum(lens), len(lens))
    n = args.reads
    lowq = args.genome == "hg19like"  # the round-1 workload keeps its clean reads
    out = None
    extra = []

    # ---------------------------------------------------------------- headline leg
    if args.mode == "se":
        strands = walt_amd.STRANDS_GA if args.ag else walt_amd.STRANDS_CT
        t0 = time.perf_counter()
        idx = walt_amd.Index.build_device(genome_ascii.data_ptr(), lens, names, device=local, strands=strands,
                                          dir_bits=args.dir_bits)
        apply_opts(idx, args)
        t_index = time.perf_counter() - t0
        s0 = 2 if args.ag else 0
        log("index: %d + %d entries, dir_bits %d, %.1f GB in HBM, outliers %d/%d, bad buckets %d/%d (%.1f s)" % (
            idx.index_size(s0), idx.index_size(s0 + 1), idx.dir_bits, idx.device_bytes / 1e9, idx.outliers(s0),
            idx.outliers(s0 + 1), idx.bad_buckets(s0), idx.bad_buckets(s0 + 1), t_index))
        t0 = time.perf_counter()
        d_bases, d_off = synth.make_reads(torch, dev, genome_ascii, n, args.read_len, seed=1000 + rank + args.seed_offset, ag=args.ag, lowq=lowq)
        torch.cuda.synchronize()
        log("reads: %d x %d bp (%.1f s)" % (n, args.read_len, time.perf_counter() - t0))
        if os.environ.get("WALT_AMD_TWICE"):  # diagnostic library only: parts of k_se_stage run twice (bit mask)
            walt_amd.lib().walt_profile_stage_stamps(int(os.environ["WALT_AMD_TWICE"]) << 8, None)
        if os.environ.get("WALT_AMD_STAMPS") == "5":  # diagnostic library only: phase sums of k_map_se_literal
            walt_amd.lib().walt_profile_stage_stamps(2, None)
        if os.environ.get("WALT_AMD_STAMPS") == "4":  # diagnostic library only: phase sums of k_se_stage
            walt_amd.lib().walt_profile_stage_stamps(1, None)
        leg = se_leg(cx, idx, d_bases, d_off, n, args.read_len, args.max_mismatches, args.bucket, args.ag, args.steps,
                     args.warmup)
        if os.environ.get("WALT_AMD_STAMPS") == "5":
            buf = (ctypes.c_ulonglong * 16)()
            walt_amd.lib().walt_profile_stage_stamps(0, buf)
            tot = float(buf[8]) or 1.0
            nm = ["read record", "care", "filter", "lookup", "masks", "candidate list", "large regions", "store"]
            log("k_map_se_literal phase shares (s_memtime, drained at boundaries): " +
                ", ".join("%s %.1f%%" % (x, 100.0 * buf[i] / tot) for i, x in enumerate(nm)))
        if os.environ.get("WALT_AMD_STAMPS") == "4":
            buf = (ctypes.c_ulonglong * 16)()
            walt_amd.lib().walt_profile_stage_stamps(0, buf)
            tot = float(buf[8]) or 1.0
            nm = ["take a read", "query + filter + directory", "danger test", "entries + resolve", "masks + small regions",
                  "dense range + mid regions", "work items", "store / lists / finished"]
            log("k_se_stage phase shares (s_memtime, drained at boundaries): " +
                ", ".join("%s %.1f%%" % (x, 100.0 * buf[i] / tot) for i, x in enumerate(nm)))
            log("k_se_stage events (all steps): fence rounds %d with %d lanes (%d searches of 128), seed steps %d with %d lanes, "
                "candidate turns %d with %d candidates" % (buf[9], buf[10], buf[15], buf[11], buf[12], buf[13], buf[14]))
        elapsed = wdist.allreduce_max(leg["elapsed"], device=dev if not shared_gpu else "cpu")  # MAX over ranks
        leg["kern_ms_ranks_max"] = wdist.allreduce_max(float(np.median(leg["map_ms"])), device=dev if not shared_gpu else "cpu")
        # mapping statistics of the last step; the ONLY data-path collective is this final sum over ranks
        # (RCCL), mirroring StatSingleReads (mapping.hpp:94-100)
        times = leg["d_out"].view(torch.int32).view(n, 4)[:, 1]
        vec = wdist.se_stats_vector(times, int(leg["stats"][0]))
        vec_local = vec.cpu().numpy().astype(np.uint64)
        if shared_gpu:
            vec = vec.cpu()
        st = wdist.allreduce_stats(vec)
        total, uniq, amb, unm, short = [int(v) for v in st.tolist()]
        c_abi = None
        if world > 1 and not shared_gpu:  # the same sum through the C ABI (walt_stats_allreduce over RCCL)
            c_abi = wdist.c_abi_cross_check(local, vec_local, [total, uniq, amb, unm, short])
        if os.environ.get("WALT_AMD_STAMPS"):
            buf = (ctypes.c_ulonglong * 16)()
            if walt_amd.lib().walt_profile_stamps(buf) == 0:
                tot = float(buf[8]) or 1.0
                nm = ["read record", "care loads", "bloom/bad", "lookup", "masks", "own-lane verify", "coop regions",
                      "store", "total"]
                log("phase shares (s_memtime, drained at boundaries): " +
                    ", ".join("%s %.1f%%" % (x, 100.0 * buf[i] / tot) for i, x in enumerate(nm[:8])))
                if int(os.environ.get("WALT_AMD_ABLATE", "0")) & 8:
                    log("danger-filter self-check: %d probe pairs checked, %d dangerous probes NOT flagged by the filter "
                        "(must be 0)" % (buf[14], buf[15]))
        log("heavy pass: %d reads, literal pass: %d reads; kernel ms/step: pack %.2f map %.2f (median)" % (
            leg["n_heavy"], leg["n_deferred"], float(np.median(leg["pack_ms"])), float(np.median(leg["map_ms"]))))
        log("staged rounds per chunk [blocked -> next round, dense items, gather items, giants]: %s" % leg["rounds"])
        log("device counters per step: probes %.2f, candidates %.2f per read, %d wave-cooperative regions" % (
            leg["stats"][1] / n, leg["stats"][2] / n, leg["stats"][3]))
        if rank == 0:
            out = {
                "metric": "mapped reads/sec (%d bp single-end%s, hg19-scale index, -m %d -b %d)%s" % (
                    args.read_len, " -A" if args.ag else "", args.max_mismatches, args.bucket,
                    "" if args.pattern == 3 else ", seed pattern %d" % args.pattern),
                "value": world * n * args.steps / elapsed, "unit": "reads/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                "config": {"workload": "configs[%d]: %s, %d x %d bp single-end %s reads per GPU, -m %d -b %d" % (
                               1 if world == 1 else 3, genome_desc, n, args.read_len, "G->A" if args.ag else "C->T",
                               args.max_mismatches, args.bucket),
                           "genome": args.genome, "genome_bp": int(sum(lens)), "sequences": len(lens), "reads_per_gpu": n,
                           "read_len": args.read_len, "max_mismatches": args.max_mismatches, "bucket_cap": args.bucket,
                           "reads": "both strands, 95 % C->T, 1 % substitutions" + (
                               "; %d %% of the reads carry %d-%d %% errors (quality tail)" % (
                                   100 * synth.LOWQ_SHARE, 100 * synth.LOWQ_LO, 100 * synth.LOWQ_HI) if lowq else ""),
                           "index_dir_bits": idx.dir_bits, "index_hbm_gb": round(idx.device_bytes / 1e9, 2),
                           "index_build_s": round(t_index, 1), "parallelism": "replica-per-gpu x%d" % world},
                "mapping": {"total": total, "unique": uniq, "ambiguous": amb, "unmapped": unm, "too_short": short,
                            "unique_frac": uniq / max(1, total)},
                "kernel_ms": {"pack_reads": float(np.median(leg["pack_ms"])), "map_se": float(np.median(leg["map_ms"])),
                              "map_se_min": float(np.min(leg["map_ms"])), "map_se_max": float(np.max(leg["map_ms"])),
                              "by_group": dict(zip(("pass1", "heavy_stages", "se_verify", "literal"), leg["detail_ms"]))},
                "device_counters_per_read": {"probes": float(leg["stats"][1]) / n, "candidates": float(leg["stats"][2]) / n,
                                             "wave_cooperative_regions_per_step": int(leg["stats"][3]),
                                             "heavy_pass_reads": int(leg["n_heavy"]),
                                             "deferred_to_literal_pass": int(leg["n_deferred"])},
            }
            if c_abi:
                out["stats_allreduce_c_abi"] = c_abi
        jobs = []
        e2e_keep = {}
        if run_cpu:
            uni, hard = se_sample(cx, leg, n, args.cpu_sample, args.hard_sample)
            sel = torch.cat([uni, hard])
            jobs.append({"bases": d_bases.view(n, args.read_len)[sel].cpu().numpy().reshape(-1), "m": int(sel.numel()),
                         "read_len": args.read_len, "max_mm": args.max_mismatches, "b": args.bucket, "ag": args.ag,
                         "timed_first": int(uni.numel()), "sel": sel, "leg": leg, "n": n, "nu": int(uni.numel()),
                         "traffic_key": "se%d%s" % (args.read_len, "ag" if args.ag else ""), "kernel": "single-end mapping kernels: k_map_se<%d> pass 1 + heavy stages, k_se_verify, k_map_se_literal" % (7 if args.read_len <= 112 else 10),
                         "target": out})
        # ---- extra leg on the same index: 150 bp single-end at -m 10 (configs[4]'s read length on the C->T side)
        leg150 = None
        want_extra = rank == 0 and world == 1 and not args.no_extra and args.pattern == 3
        if want_extra and not args.ag and args.read_len != 150:
            n150 = max(1, n // 2)
            b150, o150 = synth.make_reads(torch, dev, genome_ascii, n150, 150, seed=3000, lowq=lowq)
            leg150 = se_leg(cx, idx, b150, o150, n150, 150, 10, args.bucket, False, args.extra_steps, 1, timed_barrier=False)
            line = {"metric": "mapped reads/sec (150 bp single-end, hg19-scale index, -m 10 -b %d)" % args.bucket,
                    "value": n150 * args.extra_steps / leg150["elapsed"], "unit": "reads/s", "n_gpus": 1,
                    "steps": args.extra_steps, "warmup": 1, "ms_per_step": 1e3 * leg150["elapsed"] / args.extra_steps,
                    "dtype": "u32", "data": "synthetic",
                    "config": {"workload": "150 bp single-end C->T reads, -m 10 (read length and -m of configs[4]) on the "
                                           "headline leg's index, %d reads" % n150}}
            extra.append(line)
            if run_cpu:
                uni, hard = se_sample(cx, leg150, n150, max(1, args.cpu_sample // 4), max(1, args.hard_sample // 4))
                sel = torch.cat([uni, hard])
                jobs.append({"bases": b150.view(n150, 150)[sel].cpu().numpy().reshape(-1), "m": int(sel.numel()),
                             "read_len": 150, "max_mm": 10, "b": args.bucket, "ag": False, "timed_first": int(uni.numel()),
                             "sel": sel, "leg": leg150, "n": n150, "nu": int(uni.numel()), "traffic_key": "se150",
                             "kernel": "single-end mapping kernels: k_map_se<10> pass 1 + heavy stages, k_se_verify, k_map_se_literal", "target": line})
            del b150, o150
        if run_cpu and jobs:
            t0 = time.perf_counter()
            oracle_se_jobs(cx, idx, jobs, lens, (s0, s0 + 1))
            log("oracle pass over %s sampled reads: %.1f s" % ([j["m"] for j in jobs], time.perf_counter() - t0))
            for j in jobs:
                cpu, roof = se_report(cx, args, j["leg"], j, j["n"], j["read_len"], j["max_mm"], j["b"], j["sel"], j["nu"],
                                      j["traffic_key"], j["kernel"])
                j["target"]["roofline"] = roof
                j["target"]["cpu_baseline"] = cpu
            if args.ref_sample > 0 and args.pattern == 3 and not args.ag and world == 1:
                # the real reference binary beside it (slower than the port: it also reads its index files)
                n_e2e = min(args.e2e_reads, n) if (not args.no_extra and not args.contigs) else 0
                try:
                    out["cpu_baseline"]["reference_binary"] = reference_binary_leg(
                        cx, idx, d_bases, leg["d_out"], args.read_len, min(args.ref_sample, n), args.max_mismatches,
                        args.bucket, jobs[0]["cores"], keep=e2e_keep if n_e2e else None,
                        extra_bytes=n_e2e * (2 * args.read_len + 24 + 3 * args.read_len))
                except Exception as e:  # never let the extra leg break the bench line
                    out["cpu_baseline"]["reference_binary"] = {"skipped": "%s: %s" % (type(e).__name__, e)}
                if "scratch" in e2e_keep:
                    e2e_keep.update({"n": n_e2e, "bases": d_bases[:n_e2e * args.read_len].cpu().numpy(), "cores": jobs[0]["cores"],
                                     "times": leg["d_out"].view(torch.int32).view(n, 4)[:n_e2e, 1].cpu().numpy()})
        j = None  # (the loop variable above still referred to the last job: its leg's workspace and result tensors)
        win_cov = {"window_records": [int(idx.window_entries(s0)), int(idx.window_entries(s0 + 1))],
                   "window_eligible": [int(idx.window_eligible(s0)), int(idx.window_eligible(s0 + 1))]}
        if rank == 0 and out is not None:
            # dense candidate windows: records that exist / index slots in runs that qualify for one (a memory budget
            # that ends early truncates them: the verifier then gathers, same results, slower)
            out["config"]["dense_window_coverage"] = dict(win_cov, fraction=[
                (r / e if e else 1.0) for r, e in zip(win_cov["window_records"], win_cov["window_eligible"])])
        del jobs, leg, leg150, d_bases, d_off
        idx.close()
        torch.cuda.empty_cache()
        if rank == 0 and world == 1 and out is not None and "roofline" in out and not args.no_extra:
            g = gather_ceiling()
            out["roofline"]["gather_ceiling_lines_per_s"] = g if isinstance(g, float) else None
            if not isinstance(g, float):
                out["roofline"]["gather_ceiling_skipped"] = g["skipped"]
            elif out["roofline"].get("gather_lines_per_s"):
                out["roofline"]["gather_frac_of_ceiling"] = out["roofline"]["gather_lines_per_s"] / g
        if rank == 0 and out is not None and "cpu_baseline" in out:
            rb = out["cpu_baseline"].get("reference_binary") or {}
            out["cpu_baseline"]["reference_reads_per_s"] = rb.get("value")   # the real `walt -t N` on this box in this run
            out["cpu_baseline"]["reference_cores"] = rb.get("cores")
            out["cpu_baseline"]["reference_mapstats_equal_gpu"] = rb.get("mapstats_equal_gpu")
        if "scratch" in e2e_keep:  # the GPU is free now: the command-line binary opens its own index
            import shutil
            try:
                extra.append(e2e_leg(cx, e2e_keep["scratch"], e2e_keep["dbi"], e2e_keep["bases"], e2e_keep["times"], args.read_len,
                                     e2e_keep["n"], args.max_mismatches, args.bucket, e2e_keep["cores"], args.e2e_batch))
                out["e2e_reads_per_s"] = extra[-1].get("value")
                out["cpu_baseline"]["e2e_reads_per_s"] = extra[-1].get("value")
                out["cpu_baseline"]["e2e_reads_per_s_whole_wall"] = extra[-1].get("reads_per_s_whole_wall")
            except Exception as e:
                extra.append({"metric": "reads/s end to end", "skipped": "%s: %s" % (type(e).__name__, e)})
            finally:
                shutil.rmtree(e2e_keep["scratch"], ignore_errors=True)
            e2e_keep.clear()
    # ---------------------------------------------------------------- paired-end legs (headline with --mode pe, else extra)
    pe_cfgs = []
    if args.mode == "pe":
        pe_cfgs.append(("headline", n, args.read_len, args.max_mismatches, args.pbat, args.steps, args.warmup))
    elif rank == 0 and world == 1 and not args.no_extra and args.pattern == 3:
        npe = args.extra_pairs or n
        pe_cfgs.append(("configs[2]", npe, 100, 6, False, args.extra_steps, 1))
        pe_cfgs.append(("configs[4]", max(1, npe // 2), 150, 10, True, args.extra_steps, 1))
    if pe_cfgs:
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        free_b, total_b = torch.cuda.mem_get_info()
        log("before the 4-strand index: %.1f GB free of %.1f (torch holds %.1f GB)" % (free_b / 1e9, total_b / 1e9, torch.cuda.memory_allocated() / 1e9))
        t0 = time.perf_counter()
        idx = walt_amd.Index.build_device(genome_ascii.data_ptr(), lens, names, device=local,
                                          strands=walt_amd.STRANDS_ALL, dir_bits=args.dir_bits)
        apply_opts(idx, args)
        t_index4 = time.perf_counter() - t0
        log("index (4 strands): %.1f GB in HBM, dir_bits %d (%.1f s)" % (idx.device_bytes / 1e9, idx.dir_bits, t_index4))
        jobs = []
        se_jobs = []
        for tag, npairs, rl, mm, pbat, steps, warm in pe_cfgs:
            d1, d2, d_off = synth.make_pairs(torch, dev, genome_ascii, npairs, rl, seed=2000 + rank + args.seed_offset + rl, lowq=lowq)
            if pbat:
                # a PBAT library's mate 1 is the A-rich read: the user's (-1, -2) files are what a directional
                # library would call (mate 2, mate 1); bin/walt -P exchanges them for the mapping call, so does this
                user1, user2 = d2, d1
                d1, d2 = user2, user1
            torch.cuda.synchronize()
            headline = tag == "headline"
            leg = pe_leg(cx, idx, d1, d2, d_off, npairs, rl, mm, args.bucket, args.top_k, args.frag_range, steps, warm,
                         timed_barrier=headline)
            elapsed = wdist.allreduce_max(leg["elapsed"], device=dev if not shared_gpu else "cpu") if headline else leg["elapsed"]
            if headline:
                leg["step_s_ranks_max"] = wdist.allreduce_max(float(np.median(leg["per_step"])), device=dev if not shared_gpu else "cpu")
            log("%s paired-end 2 x %d bp: %.1f ms/step; lists %s" % (tag, rl, 1e3 * elapsed / steps, leg["lists"]))
            log("  device counters per pair: probes %.2f, candidates verified %.1f, large regions %.3f" % (
                (float(leg["stats"][1]) + float(leg["stats"][5])) / npairs, (float(leg["stats"][2]) + float(leg["stats"][6])) / npairs,
                (float(leg["stats"][3]) + float(leg["stats"][7])) / npairs))
            words = leg["d_out"].view(torch.int32).view(npairs, 16)
            vec = wdist.pe_stats_vector(words, args.frag_range, int(leg["stats"][0]), int(leg["stats"][4]))
            vec_local = vec.cpu().numpy().astype(np.uint64)
            if headline:
                if shared_gpu:
                    vec = vec.cpu()
                vec = wdist.allreduce_stats(vec)
            v = [int(x) for x in vec.tolist()]
            hist = np.array(v[14:], dtype=np.float64)
            line = {"metric": "mapped read pairs/sec (2 x %d bp paired-end%s, hg19-scale index, -m %d -k %d -L %d)" % (
                        rl, " PBAT (-P)" if pbat else "", mm, args.top_k, args.frag_range),
                    "value": (world if headline else 1) * npairs * steps / elapsed, "unit": "pairs/s",
                    "n_gpus": world if headline else 1, "steps": steps, "warmup": warm,
                    "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "u32", "data": "synthetic",
                    "config": {"workload": "%s: %s, %d pairs 2 x %d bp per GPU, fragment U[%d,500], -m %d -b %d -k %d -L %d%s" % (
                                   tag if not headline else "configs[2]", genome_desc, npairs, rl, max(120, rl), mm, args.bucket,
                                   args.top_k, args.frag_range, ", PBAT (-P): mates exchanged for mapping" if pbat else ""),
                               "genome": args.genome, "genome_bp": int(sum(lens)), "pairs_per_gpu": npairs,
                               "index_hbm_gb": round(idx.device_bytes / 1e9, 2), "index_dir_bits": idx.dir_bits},
                    "mapping": {"pairs": v[0], "unique_pairs": v[1], "ambiguous_pairs": v[2], "unpaired": v[3],
                                "mate1": dict(zip(wdist.SE_FIELDS, v[4:9])), "mate2": dict(zip(wdist.SE_FIELDS, v[9:14])),
                                "frag_len_mean": float((hist * np.arange(hist.size)).sum() / max(1.0, hist.sum())),
                                "unique_frac": v[1] / max(1, v[0])},
                    "lists_last_pass": leg["lists"]}
            if headline and world > 1 and not shared_gpu:
                line["stats_allreduce_c_abi"] = wdist.c_abi_cross_check(local, vec_local, v)
            if headline:
                out = line if rank == 0 else None
            else:
                extra.append(line)
            if run_cpu:
                uni, hard = pe_sample(cx, leg, npairs, max(1, args.cpu_sample // 10), max(1, args.hard_sample // 5))
                sel = torch.cat([uni, hard])
                jobs.append({"m1": d1.view(npairs, rl)[sel].cpu().numpy().reshape(-1),
                             "m2": d2.view(npairs, rl)[sel].cpu().numpy().reshape(-1), "m": int(sel.numel()), "read_len": rl,
                             "max_mm": mm, "b": args.bucket, "top_k": args.top_k, "frag_range": args.frag_range, "sel": sel,
                             "leg": leg, "n": npairs, "nu": int(uni.numel()), "target": line,
                             "traffic_key": "pe%d" % rl})
            del d1, d2, d_off
            torch.cuda.empty_cache()
        # ---- configs[4]'s single-end half: 150 bp A-rich reads with -A on the _GA10/_GA11 indexes, -m 10
        if args.mode != "pe" and len(pe_cfgs) > 1:
            n150 = max(1, n // 2)
            b150, o150 = synth.make_reads(torch, dev, genome_ascii, n150, 150, seed=4000, ag=True, lowq=lowq)
            legA = se_leg(cx, idx, b150, o150, n150, 150, 10, args.bucket, True, args.extra_steps, 1, timed_barrier=False)
            tA = legA["d_out"].view(torch.int32).view(n150, 4)[:, 1]
            lineA = {"metric": "mapped reads/sec (150 bp single-end -A, hg19-scale index, -m 10 -b %d)" % args.bucket,
                     "value": n150 * args.extra_steps / legA["elapsed"], "unit": "reads/s", "n_gpus": 1,
                     "steps": args.extra_steps, "warmup": 1, "ms_per_step": 1e3 * legA["elapsed"] / args.extra_steps,
                     "dtype": "u32", "data": "synthetic",
                     "config": {"workload": "configs[4] single-end half: %d x 150 bp A-rich (G->A) reads, -A -m 10, on the "
                                            "_GA10/_GA11 strands of the 4-strand index" % n150},
                     "mapping": {"unique_frac": float((tA == 1).float().mean().item())}}
            extra.append(lineA)
            if run_cpu:
                uni, hard = se_sample(cx, legA, n150, max(1, args.cpu_sample // 4), max(1, args.hard_sample // 4))
                sel = torch.cat([uni, hard])
                se_jobs.append({"bases": b150.view(n150, 150)[sel].cpu().numpy().reshape(-1), "m": int(sel.numel()),
                                "read_len": 150, "max_mm": 10, "b": args.bucket, "ag": True, "timed_first": int(uni.numel()),
                                "sel": sel, "leg": legA, "n": n150, "nu": int(uni.numel()), "traffic_key": "se150ag",
                                "kernel": "single-end mapping kernels: k_map_se<10> pass 1 + heavy stages, k_se_verify, k_map_se_literal", "target": lineA})
            del b150, o150
        if run_cpu and jobs:
            t0 = time.perf_counter()
            oracle_pe_jobs(cx, idx, jobs, lens)
            log("oracle paired-end pass over %s sampled pairs: %.1f s" % ([j["m"] for j in jobs], time.perf_counter() - t0))
            for j in jobs:
                cpu, roof = pe_report(cx, args, j["leg"], j, j["n"], j["read_len"], j["sel"], j["nu"], j["traffic_key"])
                j["target"]["cpu_baseline"] = cpu
                j["target"]["roofline"] = roof
        if run_cpu and se_jobs:
            oracle_se_jobs(cx, idx, se_jobs, lens, (2, 3))
            for j in se_jobs:
                cpu, roof = se_report(cx, args, j["leg"], j, j["n"], j["read_len"], j["max_mm"], j["b"], j["sel"], j["nu"],
                                      j["traffic_key"], j["kernel"])
                j["target"]["roofline"] = roof
                j["target"]["cpu_baseline"] = cpu
        idx.close()
    # ---------------------------------------------------------------- many sequences: the same genome cut into 3,000 contigs
    # (an assembly like hg38's analysis set has 3,366 sequences; every chromosome end adds entries the key search may
    # not trust): the single-end leg on it, every round
    if rank == 0 and world == 1 and not args.no_extra and args.pattern == 3 and args.mode == "se" and not args.contigs \
            and not args.ag and args.extra_contigs > 0 and out is not None:
        import gc
        genome_ascii = None
        gc.collect()
        torch.cuda.empty_cache()
        try:
            t0 = time.perf_counter()
            g3, lens3, names3 = synth.make_genome(torch, dev, scale, seed=2, kind=args.genome, contigs=args.extra_contigs)
            idx = walt_amd.Index.build_device(g3.data_ptr(), lens3, names3, device=local, strands=walt_amd.STRANDS_CT,
                                              dir_bits=args.dir_bits)
            apply_opts(idx, args)
            b3, o3 = synth.make_reads(torch, dev, g3, n, args.read_len, seed=5000, lowq=lowq)
            torch.cuda.synchronize()
            log("contigs leg: %d sequences, index %.1f GB, outliers %d/%d (%.1f s)" % (
                len(lens3), idx.device_bytes / 1e9, idx.outliers(0), idx.outliers(1), time.perf_counter() - t0))
            leg3 = se_leg(cx, idx, b3, o3, n, args.read_len, args.max_mismatches, args.bucket, False, args.extra_steps, 1,
                          timed_barrier=False)
            log("contigs leg: heavy pass %d reads, literal pass %d reads; map %.2f ms (median), groups %s" % (
                leg3["n_heavy"], leg3["n_deferred"], float(np.median(leg3["map_ms"])), [round(x, 2) for x in leg3["detail_ms"]]))
            line3 = {"metric": "mapped reads/sec (%d bp single-end, hg19-scale genome in %d sequences, -m %d -b %d)" % (
                         args.read_len, len(lens3), args.max_mismatches, args.bucket),
                     "value": n * args.extra_steps / leg3["elapsed"], "unit": "reads/s", "n_gpus": 1, "steps": args.extra_steps,
                     "warmup": 1, "ms_per_step": 1e3 * leg3["elapsed"] / args.extra_steps, "dtype": "u32", "data": "synthetic",
                     "config": {"workload": "contigs%d: the headline genome cut into %d equal sequences, %d x %d bp single-end "
                                            "reads, -m %d -b %d" % (args.extra_contigs, len(lens3), n, args.read_len,
                                                                    args.max_mismatches, args.bucket),
                                "sequences": len(lens3), "outliers": [int(idx.outliers(0)), int(idx.outliers(1))]},
                     "kernel_ms": {"map_se": float(np.median(leg3["map_ms"])),
                                   "by_group": dict(zip(("pass1", "heavy_stages", "se_verify", "literal"), leg3["detail_ms"]))},
                     "heavy_pass_reads": int(leg3["n_heavy"]), "deferred_to_literal_pass": int(leg3["n_deferred"])}
            if run_cpu:
                uni, hard = se_sample(cx, leg3, n, max(1, args.cpu_sample // 5), max(1, args.hard_sample // 5))
                sel = torch.cat([uni, hard])
                j3 = {"bases": b3.view(n, args.read_len)[sel].cpu().numpy().reshape(-1), "m": int(sel.numel()),
                      "read_len": args.read_len, "max_mm": args.max_mismatches, "b": args.bucket, "ag": False,
                      "timed_first": int(uni.numel())}
                oracle_se_jobs(cx, idx, [j3], lens3, (0, 1))
                cx_bp, cx.genome_bp = cx.genome_bp, -1  # (no filed PMC traffic belongs to this genome)
                cpu3, roof3 = se_report(cx, args, leg3, j3, n, args.read_len, args.max_mismatches, args.bucket, sel,
                                        int(uni.numel()), "se%d_contigs" % args.read_len,
                                        "single-end mapping kernels on %d sequences" % len(lens3))
                cx.genome_bp = cx_bp
                line3["roofline"], line3["cpu_baseline"] = roof3, cpu3
                del j3
            extra.append(line3)
            out["contigs%d_ms_per_step" % args.extra_contigs] = line3["ms_per_step"]
            if "roofline" in out:
                out["roofline"]["contigs%d_ms_per_step" % args.extra_contigs] = line3["ms_per_step"]
                out["roofline"]["contigs%d_bit_exact" % args.extra_contigs] = (line3.get("cpu_baseline") or {}).get("bit_exact_vs_gpu")
            del leg3, b3, o3, g3
            idx.close()
            torch.cuda.empty_cache()
        except Exception as e:  # never let the extra leg break the bench line
            extra.append({"metric": "contigs leg", "skipped": "%s: %s" % (type(e).__name__, e)})
    if rank == 0 and world == 1 and run_cpu and not args.no_calibration and not args.no_extra and args.pattern == 3 \
            and out is not None and "cpu_baseline" in out:
        try:
            genome_ascii = None
            torch.cuda.empty_cache()
            out["cpu_baseline"]["calibration"] = calibration_leg(cx, local, walt_amd.effective_cpus())
        except Exception as e:  # never let the extra leg break the bench line
            out["cpu_baseline"]["calibration"] = {"skipped": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        if extra:
            out["extra_lines"] = extra
            # the other legs' headline figures as scalars of the main line's roofline object (the driver's record keeps
            # scalars; the full lines are in extra_lines)
            if "roofline" in out:
                for ln in extra:
                    m = ln.get("metric", "")
                    tag = ("pe100" if "2 x 100" in m else "pe150" if "2 x 150" in m else "se150ag" if "150 bp single-end -A" in m
                           else "se150" if "150 bp single-end" in m else None)
                    if tag is None or "ms_per_step" not in ln:
                        continue
                    out["roofline"][tag + "_ms_per_step"] = ln["ms_per_step"]
                    out["roofline"][tag + "_frac"] = (ln.get("roofline") or {}).get("frac")
                    out["roofline"][tag + "_frac_lines"] = (ln.get("roofline") or {}).get("frac_lines")
                    out["roofline"][tag + "_bit_exact"] = (ln.get("cpu_baseline") or {}).get("bit_exact_vs_gpu")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()  # ranks > 0 wait for rank 0 before tearing the group down
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    if os.environ.get("WALT_AMD_BENCH_DRYRUN"):
        dryrun(args)
        return
    worker(args)


if __name__ == "__main__":
    main()
