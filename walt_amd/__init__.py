"""walt_amd -- MI355X-native seed-and-extend hot path of WALT behind a C ABI.

This package is only the thin Python binding of ``lib/libwalt_amd.so`` (built
from ``csrc/`` by ``python -m walt_amd.build`` / ``__graft_entry__.build()``);
the product is the HIP library declared in ``include/walt_amd.h``.  There is
no CPU fallback: if the library is missing the import fails loudly, and the
mapping calls return WALT_EHIP when no GPU is present.

Function names and argument meaning follow the reference call sites they
replace (smithlabcode/walt v1.0): ``map_se_batch`` is the strand loop + OpenMP
loop over ``SingleEndMapping`` (mapping.cpp:486-500), ``map_pe_batch`` the
loops over ``PairEndMapping`` plus ``MergePairedEndResults``
(paired.cpp:642-699), ``Index`` is ``ReadIndexHeadInfo``/``ReadIndex``
(reference.cpp:324-417), ``makedb`` is makedb.cpp:128-159.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# Seed pattern: a compile-time choice of the library, like the reference's -D SEEDPATTERN3 / 5 / 7
# (src/walt/Makefile:34, FAQ.md:5-13): libwalt_amd.so is pattern 3 (the default), libwalt_amd_sp5.so /
# libwalt_amd_sp7.so the other two.  set_pattern() (or WALT_AMD_PATTERN) selects the library that lib(),
# makedb() and the Index constructors use; an Index keeps the library it was made with.
PATTERN = int(os.environ.get("WALT_AMD_PATTERN", "3"))


def set_pattern(p):
    global PATTERN
    if p not in (3, 5, 7):
        raise ValueError("seed pattern must be 3, 5 or 7")
    PATTERN = p


def lib_path(pattern=None):
    pattern = PATTERN if pattern is None else pattern
    # WALT_AMD_LIB: another build of the default library (A/B timing of two builds on one GPU box; diagnostic)
    if pattern == 3 and os.environ.get("WALT_AMD_LIB"):
        return os.environ["WALT_AMD_LIB"]
    return os.path.join(_HERE, "lib", "libwalt_amd%s.so" % ("" if pattern == 3 else "_sp%d" % pattern))


LIB_PATH = lib_path(3)

WALT_OK = 0
STRAND_CT00, STRAND_CT01, STRAND_GA10, STRAND_GA11 = 1, 2, 4, 8
STRANDS_CT, STRANDS_GA, STRANDS_ALL = 3, 12, 15


def effective_cpus():
    """CPUs this process may use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n

# numpy views of the C structs (include/walt_amd.h)
best_match_dtype = np.dtype(
    [("genome_pos", "<u4"), ("times", "<u4"), ("strand", "S1"), ("pad", "V3"), ("mismatch", "<u4")])
candidate_dtype = np.dtype([("genome_pos", "<u4"), ("strand", "S1"), ("pad", "V3"), ("mismatch", "<u4")])
pair_result_dtype = np.dtype(
    [("m1", best_match_dtype), ("m2", best_match_dtype), ("best_times", "<u4"), ("frag_len", "<i4"),
     ("best_i", "<i4"), ("best_j", "<i4"), ("pair_mm", "<u4"), ("pad", "V12")])
batch_stats_dtype = np.dtype(
    [("too_short", "<u8"), ("probes", "<u8"), ("candidates", "<u8"), ("big_regions", "<u8")])
assert best_match_dtype.itemsize == 16 and candidate_dtype.itemsize == 12
assert pair_result_dtype.itemsize == 64 and batch_stats_dtype.itemsize == 32


# status codes of include/walt_amd.h
WALT_OK, WALT_EINVAL, WALT_EIO, WALT_EHIP, WALT_EBASE, WALT_ENOMEM, WALT_EFORMAT = 0, -1, -2, -3, -4, -5, -6


class WaltError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("walt_amd error %d: %s" % (code, msg))
        self.code = code


_libs = {}


def _preload_hip_runtime():
    """One process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64 (same SONAME as /opt/rocm's); if libwalt_amd.so pulled in the system
    copy first, a later `import torch` would load a second runtime and find no GPU.
    So when torch is installed, its bundled runtime is loaded first (by path, without
    importing torch) and libwalt_amd.so's NEEDED libamdhip64.so.7 resolves to it.
    WALT_AMD_HIP_RUNTIME=<path> overrides; empty string disables the preload."""
    import importlib.util
    path = os.environ.get("WALT_AMD_HIP_RUNTIME")
    if path is None:
        try:
            spec = importlib.util.find_spec("torch")
        except (ImportError, ValueError):
            spec = None
        if spec is not None and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                path = cand
    if path:
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def diag_lib():
    """libwalt_amd_diag.so (make -C walt_amd/csrc diag): the pattern-3 library with the in-kernel phase stamps and
    the WALT_AMD_ABLATE / WALT_AMD_STAMPS / WALT_AMD_SYNC_DEBUG switches, which the product library does not have."""
    return lib("diag")


def lib(pattern=None):
    """Load libwalt_amd.so / its _sp5 / _sp7 variant (fails loudly when it has not been built)."""
    pattern = PATTERN if pattern is None else pattern
    if pattern in _libs:
        return _libs[pattern]
    diag = pattern == "diag"
    path = os.path.join(_HERE, "lib", "libwalt_amd_diag.so") if diag else lib_path(pattern)
    if diag:
        pattern = 3
    if not os.path.exists(path):
        raise ImportError(
            "walt_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the hot path)" % path)
    _preload_hip_runtime()
    L = ctypes.CDLL(path)
    c = ctypes
    vp, u32, u64, ci = c.c_void_p, c.c_uint32, c.c_uint64, c.c_int
    L.walt_last_error.restype = c.c_char_p
    L.walt_device_count.restype = ci
    L.walt_min_read_len.restype = u32
    L.walt_max_read_len.restype = u32
    if L.walt_seed_pattern() != pattern:
        raise ImportError("walt_amd: %s was built for seed pattern %d, not %d" % (path, L.walt_seed_pattern(), pattern))
    L.walt_index_open.argtypes = [c.c_char_p, ci, c.c_uint, ci, c.POINTER(vp)]
    L.walt_index_from_host.argtypes = [u32, vp, vp, vp, vp, vp, vp, ci, ci, c.POINTER(vp)]
    L.walt_index_close.argtypes = [vp]
    L.walt_index_close.restype = None
    L.walt_index_n_chrom.argtypes = [vp]
    L.walt_index_n_chrom.restype = u32
    L.walt_index_chrom_len.argtypes = [vp, u32]
    L.walt_index_chrom_len.restype = u32
    L.walt_index_chrom_name.argtypes = [vp, u32]
    L.walt_index_chrom_name.restype = c.c_char_p
    L.walt_index_genome_len.argtypes = [vp]
    L.walt_index_genome_len.restype = u64
    L.walt_index_device_bytes.argtypes = [vp]
    L.walt_index_device_bytes.restype = u64
    L.walt_index_dir_bits.argtypes = [vp]
    L.walt_index_dir_bits.restype = ci
    L.walt_index_bad_buckets.argtypes = [vp, ci]
    L.walt_index_bad_buckets.restype = u64
    L.walt_index_outliers.argtypes = [vp, ci]
    L.walt_index_outliers.restype = u64
    L.walt_index_window_entries.argtypes = [vp, ci]
    L.walt_index_window_entries.restype = u64
    L.walt_index_window_eligible.argtypes = [vp, ci]
    L.walt_index_window_eligible.restype = u64
    L.walt_map_se_batch.argtypes = [vp, vp, vp, u32, ci, u32, u32, vp, vp]
    L.walt_se_workspace_bytes.argtypes = [u32, u32]
    L.walt_se_workspace_bytes.restype = c.c_size_t
    L.walt_map_se_batch_device.argtypes = [vp, vp, vp, u32, u32, ci, u32, u32, vp, vp, vp, c.c_size_t, vp]
    L.walt_batch_check.argtypes = [vp, vp]
    L.walt_map_pe_batch.argtypes = [vp, vp, vp, vp, vp, u32, u32, u32, u32, ci, vp, vp, vp, vp, vp, vp]
    L.walt_pe_workspace_bytes.argtypes = [u32, u32, u32]
    L.walt_pe_workspace_bytes.restype = c.c_size_t
    L.walt_pe_workspace_bytes_best.argtypes = [vp, u32, u32, u32]
    L.walt_pe_workspace_bytes_best.restype = c.c_size_t
    L.walt_map_pe_batch_device.argtypes = [vp, vp, vp, vp, vp, u32, u32, u32, u32, u32, ci, vp, vp, vp, c.c_size_t, vp]
    L.walt_index_set_option.argtypes = [vp, c.c_char_p, c.c_longlong]
    L.walt_index_get_option.argtypes = [vp, c.c_char_p, c.POINTER(c.c_longlong)]
    L.walt_makedb.argtypes = [c.c_char_p, c.c_char_p, ci]
    L.walt_index_build_device.argtypes = [vp, u32, vp, vp, ci, c.c_uint, ci, c.POINTER(vp)]
    L.walt_index_size.argtypes = [vp, ci]
    L.walt_index_size.restype = u32
    L.walt_index_export_strand.argtypes = [vp, ci, vp, vp, vp]
    L.walt_index_write.argtypes = [vp, c.c_char_p]
    L.walt_profile_enable.argtypes = [vp, ci]
    L.walt_profile_last.argtypes = [vp, c.POINTER(c.c_float), c.POINTER(c.c_float)]
    L.walt_profile_detail.argtypes = [vp, c.POINTER(c.c_float)]
    L.walt_comm_available.argtypes = []
    L.walt_comm_unique_id.argtypes = [vp]
    L.walt_comm_init.argtypes = [ci, ci, ci, vp, c.POINTER(vp)]
    L.walt_stats_allreduce.argtypes = [vp, vp, c.c_size_t]
    L.walt_comm_rank.argtypes = [vp]
    L.walt_comm_world.argtypes = [vp]
    L.walt_comm_close.argtypes = [vp]
    L.walt_comm_close.restype = None
    _libs["diag" if diag else pattern] = L
    return L


def _check(rc):
    if rc != WALT_OK:
        raise WaltError(rc, lib().walt_last_error().decode("utf-8", "replace"))


def device_count():
    return lib().walt_device_count()


def makedb(fasta_path, out_dbindex_path, threads=1):
    """makedb -c <fasta_path> -o <out_dbindex_path> (makedb.cpp:128-159)."""
    _check(lib().walt_makedb(os.fsencode(fasta_path), os.fsencode(out_dbindex_path), int(threads)))


def _ptr(a):
    return None if a is None else a.ctypes.data


def pack_reads(seqs):
    """list of str/bytes -> (bases uint8[total], offsets uint64[n+1])."""
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offsets[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    bases = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, dtype=np.uint8)
    return bases, offsets


COMM_ID_BYTES = 128


def comm_available():
    """True when librccl loads in this process (walt_comm_available; no communication)."""
    return lib().walt_comm_available() == 0


def comm_unique_id():
    """128-byte RCCL id made by rank 0 (ncclGetUniqueId); the caller hands it to the other ranks."""
    buf = (ctypes.c_ubyte * COMM_ID_BYTES)()
    _check(lib().walt_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
    return bytes(buf)


class Comm:
    """RCCL communicator of the one-process-per-GPU path; its only use is stats_allreduce()."""

    def __init__(self, device, rank, world, unique_id):
        self._L = lib()
        h = ctypes.c_void_p()
        buf = (ctypes.c_ubyte * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _check(self._L.walt_comm_init(int(device), int(rank), int(world), ctypes.cast(buf, ctypes.c_void_p),
                                      ctypes.byref(h)))
        self._h = h

    def stats_allreduce(self, vec):
        """Sum a vector of counters over the ranks (walt_stats_allreduce); returns a numpy uint64 array."""
        v = np.ascontiguousarray(np.asarray(vec), dtype=np.uint64).copy()
        _check(self._L.walt_stats_allreduce(self._h, v.ctypes.data, v.size))
        return v

    @property
    def rank(self):
        return self._L.walt_comm_rank(self._h)

    @property
    def world(self):
        return self._L.walt_comm_world(self._h)

    def close(self):
        if self._h:
            self._L.walt_comm_close(self._h)
            self._h = None


class Index:
    """Device-resident index (all selected strands stay in HBM)."""

    def __init__(self, handle, L=None):
        self._h = handle
        self._L = L if L is not None else lib()

    def _ck(self, rc):
        if rc != WALT_OK:
            raise WaltError(rc, self._L.walt_last_error().decode("utf-8", "replace"))

    @classmethod
    def open(cls, dbindex_path, device=0, strands=STRANDS_ALL, dir_bits=-1):
        h = ctypes.c_void_p()
        _check(lib().walt_index_open(os.fsencode(dbindex_path), int(device), int(strands), int(dir_bits),
                                     ctypes.byref(h)))
        return cls(h, lib())

    @classmethod
    def from_host(cls, chrom_len, genome, counter, index, chrom_names=None, device=0, dir_bits=-1, diag=False):
        """genome/counter/index: 4-lists (CT00, CT01, GA10, GA11) of numpy arrays or None.
        diag: through the diagnostic build of the library (diag_lib)."""
        n = len(chrom_len)
        cl = np.ascontiguousarray(chrom_len, dtype=np.uint32)
        names = None
        keep = []
        if chrom_names is not None:
            arr = (ctypes.c_char_p * n)(*[os.fsencode(x) for x in chrom_names])
            names = ctypes.cast(arr, ctypes.c_void_p)
            keep.append(arr)
        g = (ctypes.c_void_p * 4)()
        cn = (ctypes.c_void_p * 4)()
        ix = (ctypes.c_void_p * 4)()
        sz = (ctypes.c_uint32 * 4)()
        for s in range(4):
            if genome[s] is None:
                continue
            ga = np.ascontiguousarray(genome[s], dtype=np.uint8)
            ca = np.ascontiguousarray(counter[s], dtype=np.uint32)
            ia = np.ascontiguousarray(index[s], dtype=np.uint32)
            keep += [ga, ca, ia]
            g[s], cn[s], ix[s], sz[s] = ga.ctypes.data, ca.ctypes.data, ia.ctypes.data, ia.size
        h = ctypes.c_void_p()
        L = diag_lib() if diag else lib()
        rc = L.walt_index_from_host(n, cl.ctypes.data, names, ctypes.cast(g, ctypes.c_void_p),
                                    ctypes.cast(cn, ctypes.c_void_p), ctypes.cast(ix, ctypes.c_void_p),
                                    ctypes.cast(sz, ctypes.c_void_p), int(device), int(dir_bits), ctypes.byref(h))
        if rc != WALT_OK:
            raise WaltError(rc, L.walt_last_error().decode("utf-8", "replace"))
        return cls(h, L)

    @classmethod
    def build_device(cls, d_genome_ascii, chrom_len, chrom_names=None, device=0, strands=STRANDS_ALL,
                     dir_bits=-1):
        """GPU makedb: d_genome_ascii is an HBM address of the ACGT genome (makedb.cpp:46-85)."""
        n = len(chrom_len)
        cl = np.ascontiguousarray(chrom_len, dtype=np.uint32)
        names = None
        if chrom_names is not None:
            arr = (ctypes.c_char_p * n)(*[os.fsencode(x) for x in chrom_names])
            names = ctypes.cast(arr, ctypes.c_void_p)
        h = ctypes.c_void_p()
        _check(lib().walt_index_build_device(d_genome_ascii, n, cl.ctypes.data, names, int(device), int(strands),
                                             int(dir_bits), ctypes.byref(h)))
        return cls(h, lib())

    def index_size(self, strand):
        return self._L.walt_index_size(self._h, strand)

    def export_strand(self, strand, want_genome=True):
        """(genome bytes, counter, index) numpy arrays of a resident strand (reference.cpp:302-322 layout)."""
        g = np.empty(self.genome_len, dtype=np.uint8) if want_genome else None
        cnt = np.empty((1 << 24) + 1, dtype=np.uint32)
        ix = np.empty(self.index_size(strand), dtype=np.uint32)
        self._ck(self._L.walt_index_export_strand(self._h, strand, _ptr(g), _ptr(cnt), _ptr(ix)))
        return g, cnt, ix

    def write(self, dbindex_path):
        self._ck(self._L.walt_index_write(self._h, os.fsencode(dbindex_path)))

    def profile_enable(self, on=True):
        self._ck(self._L.walt_profile_enable(self._h, int(on)))

    def profile_last(self):
        a, b = ctypes.c_float(0), ctypes.c_float(0)
        self._ck(self._L.walt_profile_last(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def profile_detail(self):
        """ms of the last single-end call by kernel group: pass 1, heavy stages, region verifier, literal pass"""
        buf = (ctypes.c_float * 4)()
        self._ck(self._L.walt_profile_detail(self._h, buf))
        return [float(x) for x in buf]

    def close(self):
        if self._h:
            self._L.walt_index_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def n_chrom(self):
        return self._L.walt_index_n_chrom(self._h)

    @property
    def chrom_lengths(self):
        return [self._L.walt_index_chrom_len(self._h, i) for i in range(self.n_chrom)]

    @property
    def chrom_names(self):
        return [self._L.walt_index_chrom_name(self._h, i).decode() for i in range(self.n_chrom)]

    @property
    def genome_len(self):
        return self._L.walt_index_genome_len(self._h)

    @property
    def device_bytes(self):
        return self._L.walt_index_device_bytes(self._h)

    @property
    def dir_bits(self):
        return self._L.walt_index_dir_bits(self._h)

    def bad_buckets(self, strand):
        return self._L.walt_index_bad_buckets(self._h, strand)

    def outliers(self, strand):
        return self._L.walt_index_outliers(self._h, strand)

    def window_entries(self, strand):
        return self._L.walt_index_window_entries(self._h, strand)

    def window_eligible(self, strand):
        return self._L.walt_index_window_eligible(self._h, strand)

    # -- single-end -----------------------------------------------------------
    def map_se_batch(self, bases, offsets, ag_wildcard=False, max_mismatches=6, b=5000):
        """Host-buffer form.  Returns (best_match[n], stats)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        out = np.zeros(n, dtype=best_match_dtype)
        stats = np.zeros(1, dtype=batch_stats_dtype)
        self._ck(self._L.walt_map_se_batch(self._h, _ptr(bases), _ptr(offsets), n, int(bool(ag_wildcard)),
                                       int(max_mismatches), int(b), _ptr(out), _ptr(stats)))
        return out, stats[0]

    def map_se_batch_device(self, d_bases, d_offsets, n, max_read_len, d_out, d_stats, d_workspace, workspace_bytes,
                            stream=0, ag_wildcard=False, max_mismatches=6, b=5000):
        """Device-pointer form (ints are HBM addresses, stream a hipStream_t value); workspace_bytes = what
        d_workspace holds (at least se_workspace_bytes(n, max_read_len))."""
        self._ck(self._L.walt_map_se_batch_device(self._h, d_bases, d_offsets, int(n), int(max_read_len),
                                              int(bool(ag_wildcard)), int(max_mismatches), int(b), d_out, d_stats,
                                              d_workspace, int(workspace_bytes), stream))

    # -- options: tuning values and test hooks (include/walt_amd.h; the library reads no environment on the mapping path)
    def set_option(self, name, value):
        self._ck(self._L.walt_index_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = ctypes.c_longlong(0)
        self._ck(self._L.walt_index_get_option(self._h, name.encode(), ctypes.byref(v)))
        return v.value

    def pe_workspace_bytes(self, n, max_read_len, top_k):
        """What a paired-end call uses best on this index's device now (never less than the module-level minimum)."""
        return self._L.walt_pe_workspace_bytes_best(self._h, int(n), int(max_read_len), int(top_k))

    @staticmethod
    def check_batch(d_workspace, stream=0):
        """Waits for `stream`; raises if the last device-resident call on this workspace met an invalid read."""
        _check(lib().walt_batch_check(d_workspace, stream))

    def map_pe_batch_device(self, d_bases1, d_offsets1, d_bases2, d_offsets2, n, max_read_len, d_out, d_stats,
                            d_workspace, workspace_bytes, stream=0, max_mismatches=6, b=5000, top_k=50, frag_range=1000):
        self._ck(self._L.walt_map_pe_batch_device(self._h, d_bases1, d_offsets1, d_bases2, d_offsets2, int(n),
                                              int(max_read_len), int(max_mismatches), int(b), int(top_k),
                                              int(frag_range), d_out, d_stats, d_workspace, int(workspace_bytes), stream))

    # -- paired-end -----------------------------------------------------------
    def map_pe_batch(self, bases1, offsets1, bases2, offsets2, max_mismatches=6, b=5000, top_k=50,
                     frag_range=1000, want_ranked=False):
        bases1 = np.ascontiguousarray(bases1, dtype=np.uint8)
        bases2 = np.ascontiguousarray(bases2, dtype=np.uint8)
        offsets1 = np.ascontiguousarray(offsets1, dtype=np.uint64)
        offsets2 = np.ascontiguousarray(offsets2, dtype=np.uint64)
        n = offsets1.size - 1
        if offsets2.size - 1 != n:
            raise ValueError("The number of reads in paired-end files should be the same.")  # paired.cpp:673-677
        out = np.zeros(n, dtype=pair_result_dtype)
        stats = np.zeros(2, dtype=batch_stats_dtype)
        r1 = r2 = n1 = n2 = None
        if want_ranked:
            r1 = np.zeros((n, top_k), dtype=candidate_dtype)
            r2 = np.zeros((n, top_k), dtype=candidate_dtype)
            n1 = np.zeros(n, dtype=np.uint32)
            n2 = np.zeros(n, dtype=np.uint32)
        self._ck(self._L.walt_map_pe_batch(self._h, _ptr(bases1), _ptr(offsets1), _ptr(bases2), _ptr(offsets2), n,
                                       int(max_mismatches), int(b), int(top_k), int(frag_range), _ptr(out),
                                       _ptr(r1), _ptr(n1), _ptr(r2), _ptr(n2), _ptr(stats)))
        if want_ranked:
            return out, stats, (r1, n1, r2, n2)
        return out, stats
