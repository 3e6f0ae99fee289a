"""Test-side helpers: build/load the oracle and the CPU harness, read .dbindex
files, and restate the reference's FASTQ loader and SAM/MR/mapstats writers so
that oracle (and GPU) results can be compared byte-for-byte with the golden
outputs of the real reference binary.

TEST INFRASTRUCTURE ONLY -- nothing under walt_amd/ imports this.
Citations are file:line in smithlabcode/walt v1.0.
"""
import ctypes
import gzip
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
ORACLE_SRC = os.path.join(ROOT, "oracle", "walt_oracle.cpp")
ORACLE_SO = os.path.join(ROOT, "oracle", "build", "liboracle.so")
HARNESS_SO = os.path.join(HERE, "build", "libharness.so")
REF_WALT = os.path.join(ROOT, "oracle", "_ref", "walt")
REF_MAKEDB = os.path.join(ROOT, "oracle", "_ref", "makedb")

NUM_BUCKETS = 1 << 24

best_dtype = np.dtype([("genome_pos", "<u4"), ("times", "<u4"), ("strand", "S1"), ("pad", "V3"), ("mismatch", "<u4")])
cand_dtype = np.dtype([("genome_pos", "<u4"), ("strand", "S1"), ("pad", "V3"), ("mismatch", "<u4")])
pair_dtype = np.dtype([("m1", best_dtype), ("m2", best_dtype), ("best_times", "<u4"), ("frag_len", "<i4"),
                       ("best_i", "<i4"), ("best_j", "<i4"), ("pair_mm", "<u4")])
hpair_dtype = np.dtype([("m1", best_dtype), ("m2", best_dtype), ("best_times", "<u4"), ("frag_len", "<i4"),
                        ("best_i", "<i4"), ("best_j", "<i4"), ("pair_mm", "<u4"), ("pad", "V12")])
work_dtype = np.dtype([("probes", "<u8"), ("steps", "<u8"), ("cands", "<u8"), ("too_short", "<u8"), ("cands_big", "<u8")])
trace_dtype = np.dtype([("probes", "<u4"), ("cands", "<u4"), ("max_region", "<u4"), ("over_b", "<u4"), ("cands_big", "<u4"),
                        ("pad", "V12")])  # orc_trace


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


# Seed pattern the helpers below work with: 3 (the reference's default build) or 5 / 7 (the reference rebuilt
# with -D SEEDPATTERN5 / 7).  It selects the oracle / harness builds (-DORC_PAT / -DWALT_SEEDPATTERN), the
# golden case set (cases_sp5.json, out_sp5/ ...) and MINIMALREADLEN in the mapstats text.
PATTERN = 3
MIN_READ_LEN = {3: 38, 5: 32, 7: 23}
MAX_READ_LEN = {3: 1024, 5: 148, 7: 152}


def set_pattern(p):
    global PATTERN
    assert p in (3, 5, 7)
    PATTERN = p


def _sfx(p=None):
    p = PATTERN if p is None else p
    return "" if p == 3 else "_sp%d" % p


def ref_walt(p=None):
    return REF_WALT + _sfx(p)


def ref_makedb(p=None):
    return REF_MAKEDB + _sfx(p)


def build_oracle(force=False, pattern=None):
    pattern = PATTERN if pattern is None else pattern
    so = ORACLE_SO.replace(".so", _sfx(pattern) + ".so")
    if not force and _newer(so, [ORACLE_SRC]):
        return so
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(["g++", "-O3", "-fopenmp", "-shared", "-fPIC", "-std=c++11", "-DORC_PAT=%d" % pattern, "-o", so,
                    ORACLE_SRC], check=True)
    return so


def build_harness(force=False, pattern=None):
    pattern = PATTERN if pattern is None else pattern
    so = HARNESS_SO.replace(".so", _sfx(pattern) + ".so")
    csrc = os.path.join(ROOT, "walt_amd", "csrc")
    srcs = [os.path.join(HERE, "host_harness.cpp"), os.path.join(csrc, "host_index.cpp"),
            os.path.join(csrc, "core.h"), os.path.join(csrc, "index_core.h"), os.path.join(csrc, "host_common.h")]
    if not force and _newer(so, srcs):
        return so
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(["g++", "-O2", "-fopenmp", "-shared", "-fPIC", "-std=c++17", "-Wno-unknown-pragmas",
                    "-DWALT_SEEDPATTERN=%d" % pattern, "-o", so, srcs[0], srcs[1]], check=True)
    return so


HOSTIO_SO = os.path.join(HERE, "build", "libhostio_harness.so")


def build_hostio_harness(force=False):
    srcs = [os.path.join(HERE, "hostio_harness.cpp"), os.path.join(ROOT, "walt_amd", "csrc", "host", "hostio.h")]
    if not force and _newer(HOSTIO_SO, srcs):
        return HOSTIO_SO
    os.makedirs(os.path.dirname(HOSTIO_SO), exist_ok=True)
    subprocess.run(["g++", "-O2", "-fopenmp", "-shared", "-fPIC", "-std=c++17", "-Wall", "-o", HOSTIO_SO, srcs[0]],
                   check=True)
    return HOSTIO_SO


_oracle = {}
_harness = {}
_hostio = None


def hostio_harness():
    global _hostio
    if _hostio is None:
        L = ctypes.CDLL(build_hostio_harness())
        L.hio_dump.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int, ctypes.c_int,
                               ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        _hostio = L
    return _hostio


def oracle():
    if PATTERN not in _oracle:
        L = ctypes.CDLL(build_oracle())
        assert L.orc_pattern() == PATTERN
        vp, u32, ci = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int
        L.orc_get_tables.argtypes = [vp, vp]
        L.orc_hash.argtypes = [ctypes.c_char_p]
        L.orc_hash.restype = u32
        L.orc_se_map_batch.argtypes = [vp, vp, vp, u32, ci, u32, u32, ci, vp, vp]
        L.orc_se_init.argtypes = [vp, u32, u32]
        L.orc_se_map_strand.argtypes = [vp, ctypes.c_char, vp, vp, u32, ci, u32, ci, vp, vp]
        L.orc_se_map_strand_trace.argtypes = [vp, ctypes.c_char, vp, vp, u32, ci, u32, ci, vp, vp, vp]
        L.orc_pe_topk_batch.argtypes = [vp, vp, vp, u32, ci, u32, u32, u32, ci, vp, vp, vp]
        L.orc_pe_merge_batch.argtypes = [vp, vp, vp, vp, u32, vp, vp, u32, vp, u32, ci, u32, vp]
        _oracle[PATTERN] = L
    return _oracle[PATTERN]


def harness():
    if PATTERN not in _harness:
        L = ctypes.CDLL(build_harness())
        assert L.hh_pattern() == PATTERN
        vp, u32, ci = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int
        L.hh_index_new.argtypes = [u32, vp, ci]
        L.hh_index_new.restype = vp
        L.hh_index_add_strand.argtypes = [vp, ci, vp, u32, vp, vp, u32]
        L.hh_index_add_strand.restype = ctypes.c_long
        L.hh_index_force_bad.argtypes = [vp, ci, ci]
        L.hh_index_free.argtypes = [vp]
        L.hh_map_se.argtypes = [vp, vp, vp, u32, ci, u32, u32, vp, vp]
        L.hh_pe_topk.argtypes = [vp, vp, vp, u32, ci, u32, u32, u32, vp, vp, vp]
        L.hh_pe_merge.argtypes = [vp, vp, vp, vp, vp, u32, vp, vp, u32, ci, u32, vp]
        L.hh_get_nocare.argtypes = [vp]
        L.hh_region_check.argtypes = [vp, vp, vp, u32, ci, vp]
        L.hh_tail_check.argtypes = [vp, vp, vp, u32, ci, vp]
        L.hh_memo_stats.argtypes = [vp]
        L.hh_kary_check.argtypes = [u32, u32]
        L.hh_kary_check.restype = ctypes.c_long
        L.hh_fence_check.argtypes = [u32, u32]
        L.hh_fence_check.restype = ctypes.c_long
        L.hh_candidate_list_check.argtypes = [u32, u32]
        L.hh_candidate_list_check.restype = ctypes.c_long
        L.hh_tail_mask_check.argtypes = [u32]
        L.hh_tail_mask_check.restype = ctypes.c_long
        L.hh_pack.argtypes = [vp, vp, u32, ci, u32, u32, vp, ctypes.c_uint64]
        L.walt_makedb.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ci]
        L.walt_last_error.restype = ctypes.c_char_p
        _harness[PATTERN] = L
    return _harness[PATTERN]


# ---------------------------------------------------------------------------
# .dbindex reader (reference.cpp:324-351, 381-417)
# ---------------------------------------------------------------------------
STRAND_SUFFIX = ("_CT00", "_CT01", "_GA10", "_GA11")


class OrcStrand(ctypes.Structure):
    _fields_ = [("genome", ctypes.c_void_p), ("genome_len", ctypes.c_uint64), ("counter", ctypes.c_void_p),
                ("index", ctypes.c_void_p), ("index_size", ctypes.c_uint32), ("start_index", ctypes.c_void_p),
                ("n_chrom", ctypes.c_uint32)]


def make_orc_strand(genome, counter, index, start_index):
    """orc_strand over numpy arrays (kept alive by the caller)."""
    x = OrcStrand()
    x.genome = genome.ctypes.data
    x.genome_len = genome.size
    x.counter = counter.ctypes.data
    x.index = index.ctypes.data
    x.index_size = index.size
    x.start_index = start_index.ctypes.data
    x.n_chrom = start_index.size - 1
    return x


class DbIndex:
    def __init__(self, path, strands=(0, 1, 2, 3)):
        self.path = path
        with open(path, "rb") as f:
            n = int(np.frombuffer(f.read(4), "<u4")[0])
            self.names = []
            for _ in range(n):
                ln = int(np.frombuffer(f.read(4), "<u4")[0])
                self.names.append(f.read(ln).decode())
            self.lengths = np.frombuffer(f.read(4 * n), "<u4").copy()
            self.genome_len = int(np.frombuffer(f.read(4), "<u4")[0])
            self.max_index_size = int(np.frombuffer(f.read(4), "<u4")[0])
        self.n_chrom = n
        self.start_index = np.zeros(n + 1, dtype=np.uint32)
        self.start_index[1:] = np.cumsum(self.lengths, dtype=np.uint64).astype(np.uint32)
        self.genome = [None] * 4
        self.counter = [None] * 4
        self.index = [None] * 4
        for s in strands:
            with open(path + STRAND_SUFFIX[s], "rb") as f:
                f.read(1)
                self.genome[s] = np.frombuffer(f.read(self.genome_len), np.uint8).copy()
                csz, isz = np.frombuffer(f.read(8), "<u4")
                assert csz == NUM_BUCKETS
                self.counter[s] = np.frombuffer(f.read(4 * (int(csz) + 1)), "<u4").copy()
                self.index[s] = np.frombuffer(f.read(4 * int(isz)), "<u4").copy()

    def oracle_strands(self, s0):
        """ctypes array of two orc_strand structs for strands s0, s0+1."""
        arr = (OrcStrand * 2)()
        for k in range(2):
            s = s0 + k
            arr[k].genome = self.genome[s].ctypes.data
            arr[k].genome_len = self.genome_len
            arr[k].counter = self.counter[s].ctypes.data
            arr[k].index = self.index[s].ctypes.data
            arr[k].index_size = self.index[s].size
            arr[k].start_index = self.start_index.ctypes.data
            arr[k].n_chrom = self.n_chrom
        return arr

    def chrom_of(self, pos):  # getChromID, reference.cpp:43-60
        return int(np.searchsorted(self.start_index, pos, side="right") - 1) if self.n_chrom > 1 else 0


# ---------------------------------------------------------------------------
# oracle drivers
# ---------------------------------------------------------------------------
def pack_reads(seqs):
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    bases = np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8).copy()
    return bases, offsets


def oracle_se(db, seqs, ag=False, max_mm=6, b=5000, threads=4):
    bases, offsets = pack_reads(seqs)
    n = len(seqs)
    out = np.zeros(n, dtype=best_dtype)
    work = np.zeros(1, dtype=work_dtype)
    strands = db.oracle_strands(2 if ag else 0)
    oracle().orc_se_map_batch(ctypes.addressof(strands), bases.ctypes.data, offsets.ctypes.data, n, int(ag), max_mm,
                              b, threads, out.ctypes.data, work.ctypes.data)
    return out, work[0]


def oracle_pe_topk(db, seqs, ag, max_mm=6, b=5000, top_k=50, threads=4):
    bases, offsets = pack_reads(seqs)
    n = len(seqs)
    ranked = np.zeros((n, top_k), dtype=cand_dtype)
    cnt = np.zeros(n, dtype=np.uint32)
    work = np.zeros(1, dtype=work_dtype)
    strands = db.oracle_strands(2 if ag else 0)
    oracle().orc_pe_topk_batch(ctypes.addressof(strands), bases.ctypes.data, offsets.ctypes.data, n, int(ag), max_mm,
                               b, top_k, threads, ranked.ctypes.data, cnt.ctypes.data, work.ctypes.data)
    return ranked, cnt, work[0]


def oracle_pe(db, seqs1, seqs2, max_mm=6, b=5000, top_k=50, frag_range=1000, threads=4):
    r1, n1, w1 = oracle_pe_topk(db, seqs1, False, max_mm, b, top_k, threads)
    r2, n2, w2 = oracle_pe_topk(db, seqs2, True, max_mm, b, top_k, threads)
    _, off1 = pack_reads(seqs1)
    _, off2 = pack_reads(seqs2)
    n = len(seqs1)
    out = np.zeros(n, dtype=pair_dtype)
    oracle().orc_pe_merge_batch(r1.ctypes.data, n1.ctypes.data, r2.ctypes.data, n2.ctypes.data, top_k,
                                off1.ctypes.data, off2.ctypes.data, n, db.start_index.ctypes.data, db.n_chrom,
                                frag_range, max_mm, out.ctypes.data)
    return out, (r1, n1, r2, n2), (w1, w2)


# ---------------------------------------------------------------------------
# CPU harness drivers (per-lane kernel logic compiled with g++)
# ---------------------------------------------------------------------------
class HarnessIndex:
    def __init__(self, db, dir_bits, strands=(0, 1, 2, 3)):
        H = harness()
        self.db = db
        self.h = H.hh_index_new(db.n_chrom, db.lengths.ctypes.data, dir_bits)
        self.bad = {}
        for s in strands:
            if db.genome[s] is None:
                continue
            nb = H.hh_index_add_strand(self.h, s, db.genome[s].ctypes.data, db.genome_len, db.counter[s].ctypes.data,
                                       db.index[s].ctypes.data, db.index[s].size)
            assert nb >= 0, "invalid index"
            self.bad[s] = nb

    def force_bad(self, strand, on):
        harness().hh_index_force_bad(self.h, strand, int(on))

    def close(self):
        if self.h:
            harness().hh_index_free(self.h)
            self.h = None

    def map_se(self, seqs, ag=False, max_mm=6, b=5000):
        bases, offsets = pack_reads(seqs)
        n = len(seqs)
        out = np.zeros(n, dtype=best_dtype)
        ts = ctypes.c_uint64(0)
        rc = harness().hh_map_se(self.h, bases.ctypes.data, offsets.ctypes.data, n, int(ag), max_mm, b,
                                 out.ctypes.data, ctypes.addressof(ts))
        assert rc == 0, rc
        return out, ts.value

    def region_check(self, seqs, ag=False):
        """(probes, safe probes whose key-search region differs from the literal one, dangerous probes,
        safe probes sharing an outlier's prefix) -- host_harness.cpp hh_region_check"""
        bases, offsets = pack_reads(seqs)
        out = np.zeros(4, dtype=np.uint64)
        rc = harness().hh_region_check(self.h, bases.ctypes.data, offsets.ctypes.data, len(seqs), int(ag),
                                       out.ctypes.data)
        assert rc == 0, rc
        return [int(v) for v in out]

    @staticmethod
    def memo_stats():
        """counters of the memoised literal searches since the last call (host_harness.cpp hh_memo_stats): dangerous probes
        searched, of them through the memo, entry loads they made, bytes the reference's bisection reads in them"""
        out = np.zeros(70, dtype=np.uint64)
        harness().hh_memo_stats(out.ctypes.data)
        return [int(v) for v in out]

    def tail_check(self, seqs, ag=False):
        """(long-seed safe probes with a key-equal range, of them deferrable, deferrable ones whose verifier-side set
        differs from IndexRegion's, ranges holding a run breaker) -- host_harness.cpp hh_tail_check"""
        bases, offsets = pack_reads(seqs)
        out = np.zeros(4, dtype=np.uint64)
        rc = harness().hh_tail_check(self.h, bases.ctypes.data, offsets.ctypes.data, len(seqs), int(ag), out.ctypes.data)
        assert rc == 0, rc
        return [int(v) for v in out]

    def pe_topk(self, seqs, ag, max_mm=6, b=5000, top_k=50):
        bases, offsets = pack_reads(seqs)
        n = len(seqs)
        ranked = np.zeros((n, top_k), dtype=cand_dtype)
        cnt = np.zeros(n, dtype=np.uint32)
        ts = ctypes.c_uint64(0)
        rc = harness().hh_pe_topk(self.h, bases.ctypes.data, offsets.ctypes.data, n, int(ag), max_mm, b, top_k,
                                  ranked.ctypes.data, cnt.ctypes.data, ctypes.addressof(ts))
        assert rc == 0, rc
        return ranked, cnt, ts.value

    def pe_merge(self, r1, n1, r2, n2, seqs1, seqs2, top_k, frag_range=1000, max_mm=6):
        _, off1 = pack_reads(seqs1)
        _, off2 = pack_reads(seqs2)
        n = len(seqs1)
        out = np.zeros(n, dtype=hpair_dtype)
        harness().hh_pe_merge(self.h, r1.ctypes.data, n1.ctypes.data, r2.ctypes.data, n2.ctypes.data, top_k,
                              off1.ctypes.data, off2.ctypes.data, n, frag_range, max_mm, out.ctypes.data)
        return out


# ---------------------------------------------------------------------------
# FASTQ loader, LoadReadsFromFastqFile (mapping.cpp:65-121): srand(0) per batch,
# non-ACGT -> "ACGT"[rand() % 4] (util.hpp:156-163), names cut at first space.
# ---------------------------------------------------------------------------
_libc = ctypes.CDLL("libc.so.6")
_libc.rand.restype = ctypes.c_int


def _similarity(s, pos, adaptor):  # util.hpp:193-200
    lim = min(len(s) - pos, len(adaptor), 14)
    return sum(1 for i in range(lim) if s[pos + i] == adaptor[i])


def clip_adaptor(adaptor, s):
    """clip_adaptor_from_read, util.hpp:202-216 (s: bytearray, clipped tail -> 'N')."""
    n = len(s)
    lim1 = n - 14 + 1
    for i in range(max(lim1, 0)):
        if _similarity(s, i, adaptor) >= 11:
            s[i:] = b"N" * (n - i)
            return
    for i in range(max(lim1, 0), n - 5 + 1):
        if _similarity(s, i, adaptor) >= n - i - 1:
            s[i:] = b"N" * (n - i)
            return


def load_fastq_batches(path, batch_size, adaptor=""):
    """Yields (names, seqs, scores) per batch exactly as the reference loads them."""
    adaptor = adaptor.encode() if isinstance(adaptor, str) else adaptor
    with open(path, "rb") as f:
        done = False
        while not done:
            _libc.srand(0)
            names, seqs, scores = [], [], []
            name = seq = None
            line_code = 0
            line_count = 0
            lim = batch_size * 4
            while line_count < lim:
                raw = f.readline(999)  # fgets(cline, 1000, fin)
                if not raw:
                    done = True
                    break
                line = raw[:-1]  # cline[strlen(cline) - 1] = 0
                if len(line) == 0:
                    continue
                if line_code == 0:
                    sp = line.find(b" ")
                    name = line[1:] if sp <= 0 else line[1:sp]  # substr(1, 0 - 1): the unsigned wrap keeps the rest
                elif line_code == 1:
                    s = bytearray(line)
                    if adaptor:
                        clip_adaptor(adaptor, s)
                    for i, c in enumerate(s):
                        if c not in b"ACGT":
                            s[i] = b"ACGT"[_libc.rand() % 4]
                    seq = bytes(s)
                elif line_code == 3:
                    names.append(name.decode())
                    seqs.append(seq.decode())
                    scores.append(line.decode())
                line_count += 1
                line_code = (line_code + 1) % 4
            if names:
                yield names, seqs, scores
            if len(names) < batch_size:
                done = True


_COMP = str.maketrans("ACGTN", "TGCAN")


def revcomp(s):
    return s.translate(_COMP)[::-1]


def sam_header(db):  # SAMHead, reference.cpp:430-440
    out = ["@HD\tVN:1.0\n"]
    for nm, ln in zip(db.names, db.lengths):
        out.append("@SQ\tSN:%s\tLN:%u\n" % (nm, ln))
    out.append("@PG\tID:WALT\tVN:1.0\tCL:walt\n")
    return "".join(out)


def _strand(rec):
    s = rec["strand"]
    return s.decode() if isinstance(s, bytes) else s


def se_sam_line(db, rec, name, seq, score, ambiguous, unmapped):  # OutputSingleSAM, mapping.cpp:382-419
    times = int(rec["times"])
    strand = _strand(rec)
    gp = int(rec["genome_pos"])
    chr_id = db.chrom_of(gp)
    start = gp - int(db.start_index[chr_id])
    if strand == "-":
        start = (int(db.lengths[chr_id]) - start - len(seq)) & 0xFFFFFFFF
        seq, score = revcomp(seq), score[::-1]
    flag = (4 if times == 0 else 0) + (0x10 if strand == "-" else 0) + (0x100 if times >= 2 else 0)
    if times == 0 and unmapped:
        return "%s\t%d\t*\t0\t255\t*\t*\t0\t0\t%s\t%s\tNM:i:0\n" % (name, flag, seq, score)
    if times == 1 or (times >= 2 and ambiguous):
        return "%s\t%d\t%s\t%u\t255\t%uM\t*\t0\t0\t%s\t%s\tNM:i:%u\n" % (
            name, flag, db.names[chr_id], (start + 1) & 0xFFFFFFFF, len(seq), seq, score, int(rec["mismatch"]))
    return ""


def se_mr_line(db, rec, name, seq, score, ag):  # OutputUniquelyAndAmbiguousMapped, mapping.cpp:329-347
    strand = _strand(rec)
    gp = int(rec["genome_pos"])
    chr_id = db.chrom_of(gp)
    start = gp - int(db.start_index[chr_id])
    if strand == "-":
        start = (int(db.lengths[chr_id]) - start - len(seq)) & 0xFFFFFFFF
    end = (start + len(seq)) & 0xFFFFFFFF
    if ag:
        strand = "-" if strand == "+" else "+"
    return "%s\t%u\t%u\t%s\t%u\t%s\t%s\t%s\n" % (db.names[chr_id], start, end, name, int(rec["mismatch"]), strand,
                                                 seq, score)


def se_mr_route(db, rec, name, seq, score, ag, ambiguous, unmapped):
    """OutputSingleResults, mapping.cpp:355-380 -> (main, ambiguous_file, unmapped_file) text."""
    if ag:
        seq, score = revcomp(seq), score[::-1]
    times = int(rec["times"])
    if times == 0 and unmapped:
        return "", "", "%s\t%s\t%s\n" % (name, seq, score)
    if times == 1:
        return se_mr_line(db, rec, name, seq, score, ag), "", ""
    if times >= 2 and ambiguous:
        return "", se_mr_line(db, rec, name, seq, score, ag), ""
    return "", "", ""


def _pct(a, b):
    if b == 0:
        return "-nan"
    return "%g" % (100.0 * a / b)


def se_mapstats(total, unique, ambiguous, unmapped, too_short, tabs=0):  # StatSingleReads::tostring, mapping.cpp:47-63
    t = "    " * tabs
    return ("%stotal_reads: %d\n%smapped:\n%s    unique: %d\n%s    percent_unique: %s\n%s    ambiguous: %d\n"
            "%sunmapped: %d\n%smin_read_length: %d\n%stoo_short: %d") % (
                t, total, t, t, unique, t, _pct(unique, total), t, ambiguous, t, unmapped, t, MIN_READ_LEN[PATTERN], t,
                too_short)


# ---------------------------------------------------------------------------
# paired-end writers (paired.cpp:210-294, 333-435, 515-569, 52-77)
# ---------------------------------------------------------------------------
def _fwd(db, gp, strand, chr_id, read_len):  # ForwardChromPosition, paired.cpp:98-104
    s = (gp - int(db.start_index[chr_id])) & 0xFFFFFFFF
    if strand != "+":
        s = (int(db.lengths[chr_id]) - s - read_len) & 0xFFFFFFFF
    return s, (s + read_len) & 0xFFFFFFFF


def pe_frag_mr_line(db, c1, c2, name, seq1, scr1, seq2, scr2, frag_range):
    """OutputBestPairedResults MR branch, paired.cpp:210-294 -> (len, text)."""
    def i32(x):
        x &= 0xFFFFFFFF
        return x - (1 << 32) if x >= (1 << 31) else x
    seq2r, scr2r = revcomp(seq2), scr2[::-1]
    st1, st2 = _strand(c1), _strand(c2)
    g1, g2 = int(c1["genome_pos"]), int(c2["genome_pos"])
    ch1, ch2 = db.chrom_of(g1), db.chrom_of(g2)
    L1, L2 = len(seq1), len(seq2)
    s1, e1 = _fwd(db, g1, st1, ch1, L1)
    s2, e2 = _fwd(db, g2, st2, ch2, L2)
    ov_s, ov_e = max(s1, s2), min(e1, e2)
    plus = st1 == "+"
    one_l = s1 if plus else max(ov_e, s1)
    one_r = min(ov_s, e1) if plus else e1
    two_l = max(ov_e, s2) if plus else s2
    two_r = e2 if plus else min(ov_s, e2)
    ln = i32(two_r - one_l) if plus else i32(one_r - two_l)
    seq = ["N"] * max(ln, 0)
    scr = ["B"] * max(ln, 0)
    if 0 < ln <= frag_range:
        lim_one = (one_r - one_l) & 0xFFFFFFFF
        seq[:lim_one] = seq1[:lim_one]
        scr[:lim_one] = scr1[:lim_one]
        lim_two = (two_r - two_l) & 0xFFFFFFFF
        if lim_two:
            seq[ln - lim_two:] = seq2r[len(seq2r) - lim_two:]
            scr[ln - lim_two:] = scr2r[len(scr2r) - lim_two:]
        if ov_s < ov_e:
            info_one = L1 - (seq1.count("N") + int(c1["mismatch"]))
            info_two = L2 - (seq2r.count("N") + int(c2["mismatch"]))
            if info_one >= info_two:
                a = (ov_s - s1) if plus else (e1 - ov_e)
                b = (ov_e - s1) if plus else (e1 - ov_s)
                seq[lim_one:lim_one + (b - a)] = seq1[a:b]
                scr[lim_one:lim_one + (b - a)] = scr1[a:b]
            else:
                a = (ov_s - s2) if plus else (e2 - ov_e)
                b = (ov_e - s2) if plus else (e2 - ov_s)
                seq[lim_one:lim_one + (b - a)] = seq2r[a:b]
                scr[lim_one:lim_one + (b - a)] = scr2r[a:b]
    start = s1 if plus else s2
    text = "%s\t%u\t%u\tFRAG:%s\t%u\t%s\t%s\t%s\n" % (
        db.names[ch1], start, (start + ln) & 0xFFFFFFFF, name, int(c1["mismatch"]) + int(c2["mismatch"]), st1,
        "".join(seq), "".join(scr))
    return ln, text


def pe_sam_flag(paired_mapped, unmapped, next_unmapped, rev, next_rev, first, secondary):  # GetSAMFLAG, paired.cpp:80-95
    return (1 + (2 if paired_mapped else 0) + (4 if unmapped else 0) + (8 if next_unmapped else 0) +
            (0x10 if rev else 0) + (0x20 if next_rev else 0) + (0x40 if first else 0x80) +
            (0x100 if secondary else 0))


def pe_sam_lines(db, m1, m2, is_paired, ln, name, seq1, scr1, seq2, scr2, ambiguous, unmapped):
    """OutputPairedSAM, paired.cpp:333-435 (+ flags from 557-565)."""
    t1, t2 = int(m1["times"]), int(m2["times"])
    st1, st2 = _strand(m1), _strand(m2)
    f1 = pe_sam_flag(is_paired, t1 == 0, t2 == 0, st1 == "-", st2 == "-", True, t1 >= 2)
    f2 = pe_sam_flag(is_paired, t2 == 0, t1 == 0, st2 == "-", st1 == "-", False, t2 >= 2)
    g1, g2 = int(m1["genome_pos"]), int(m2["genome_pos"])
    ch1, ch2 = db.chrom_of(g1), db.chrom_of(g2)
    s1, _ = _fwd(db, g1, st1, ch1, len(seq1))
    s2, _ = _fwd(db, g2, st2, ch2, len(seq2))
    mm1, mm2 = int(m1["mismatch"]), int(m2["mismatch"])
    if t1 == 0:
        s1, mm1 = 0, 0
    else:
        s1 = (s1 + 1) & 0xFFFFFFFF
    if t2 == 0:
        s2, mm2 = 0, 0
    else:
        s2 = (s2 + 1) & 0xFFFFFFFF
    len1 = ln if st1 == "+" else -ln
    len2 = ln if st2 == "+" else -ln
    if f1 & 2:
        rn1 = rn2 = "="
    else:
        rn1 = "*" if t1 == 0 else db.names[ch1]
        rn2 = "*" if t2 == 0 else db.names[ch2]
    if st1 == "-":
        seq1, scr1 = revcomp(seq1), scr1[::-1]
    if st2 == "-":
        seq2, scr2 = revcomp(seq2), scr2[::-1]
    out = ""
    if t1 == 0 and unmapped:
        out += "%s\t%d\t*\t%u\t255\t*\t%s\t%u\t%d\t%s\t%s\tNM:i:%u\n" % (name, f1, s1, rn2, s2, len1, seq1, scr1, mm1)
    elif t1 == 1 or (t1 >= 2 and ambiguous):
        out += "%s\t%d\t%s\t%u\t255\t%uM\t%s\t%u\t%d\t%s\t%s\tNM:i:%u\n" % (
            name, f1, db.names[ch1], s1, len(seq1), rn2, s2, len1, seq1, scr1, mm1)
    if t2 == 0 and unmapped:
        out += "%s\t%d\t*\t%u\t255\t*\t%s\t%u\t%d\t%s\t%s\tNM:i:%u\n" % (name, f2, s2, rn1, s1, len2, seq2, scr2, mm2)
    elif t2 == 1 or (t2 >= 2 and ambiguous):
        out += "%s\t%d\t%s\t%u\t255\t%uM\t%s\t%u\t%d\t%s\t%s\tNM:i:%u\n" % (
            name, f2, db.names[ch2], s2, len(seq2), rn1, s1, len2, seq2, scr2, mm2)
    return out


def pe_mapstats(pairs, s1, s2, hist):  # StatPairedReads::tostring, paired.cpp:52-77
    total, uniq, amb, unm = pairs
    out = ("pairs:\n    total_read_pairs: %d\n    mapped:\n        unique: %d\n        percent_unique: %s\n"
           "        ambiguous: %d\n    unmapped: %d\nmate1:\n%s\nmate2:\n%s\n") % (
               total, uniq, _pct(uniq, total), amb, unm, se_mapstats(*s1, tabs=1), se_mapstats(*s2, tabs=1))
    out += "frag_len_distribution:\n"
    tot = 0.0
    for i, c in enumerate(hist):
        out += "    %d: %d\n" % (i, c)
        tot += i * c
    den = float(sum(hist))
    out += "frag_len_mean: " + ("%g" % (tot / den) if den else "-nan")
    return out


# ---------------------------------------------------------------------------
# golden fixtures
# ---------------------------------------------------------------------------
def golden_meta(pattern=None):
    with open(os.path.join(GOLDEN, "cases%s.json" % _sfx(pattern))) as f:
        return json.load(f)


def golden_file(case, name, pattern=None):
    with gzip.open(os.path.join(GOLDEN, "out" + _sfx(pattern), case, name + ".gz"), "rb") as f:
        return f.read().decode()


def args_to_opts(args):
    """reference CLI args (walt.cpp:130-166) -> dict with defaults (walt.cpp:103-126)."""
    o = dict(sam=False, ambiguous=False, unmapped=False, ag=False, m=6, N=10000000, b=5000, k=50, L=1000, t=1, C="")
    i = 0
    while i < len(args):
        a = args[i]
        if a == "-sam":
            o["sam"] = True
        elif a == "-a":
            o["ambiguous"] = True
        elif a == "-u":
            o["unmapped"] = True
        elif a == "-A":
            o["ag"] = True
        elif a == "-C":
            o["C"] = args[i + 1]
            i += 1
        else:
            o[a[1:]] = int(args[i + 1])
            i += 1
        i += 1
    return o
