"""Synthetic workloads for bench.py, tools/hg19_e2e.py and the GPU tests: genomes and reads generated
with torch on the device (deterministic per seed), because hg19 itself cannot travel to the GPU box.

Two genomes:
  kind="hg19like" (default of bench.py since round 2): hg19's 93 sequences (24 chromosomes, chrM, 9 haplotype
      contigs and 59 unplaced contigs; 3,137,161,264 bp) with repeat content that scales with the genome:
        * Alu-like SINE family   1.1 M x 300 bp, three age classes (2-20 % from the consensus), indels
        * L1-like LINE family    0.5 M x 0.3-6 kb, 5'-truncated copies of a 6 kb consensus (mean ~1 kb), indels
        * segmental duplications 5 % of the genome in 5-50 kb blocks copied at 0-2 % divergence
        * satellites             2.5 % of the genome in 50-500 kb tandem arrays of 171 bp-monomer units
                                 (private units: regions of hundreds of candidates; shared units: regions
                                 beyond -b 5000 and raw buckets near the 500,000 erase threshold)
        * simple repeats         0.5 M loci of 1-6 bp motifs, 20-120 bp, and poly-purine / poly-pyrimidine runs
                                 (one letter after conversion: the raw buckets makedb erases, reference.cpp:211)
      Everything else is iid bases -- which is also what makedb turns hg19's 7.6 % of N into
      (reference.cpp:123-124).  Both orientations of every family are planted.
      Calibration target (reference doc/Supplementary Data.pdf): 84.55 % of SRR1532534's reads map uniquely
      (Table S4) and, of those, 12 % were found in a region of more than one candidate, 6.5 % of more than 10,
      3.4 % of more than 100, 1.5 % of more than 1,000 and 0.7 % of more than 5,000 (Table S2).
  kind="easy": the round-1 genome (24 chromosomes, iid + four small families), kept for continuity.
"""

HG19_CHROMS = [249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022, 141213431,
               135534747, 135006516, 133851895, 115169878, 107349540, 102531392, 90354753, 81195210, 78077248, 59128983,
               63025520, 48129895, 51304566, 155270560, 59373566]
HG19_CHROM_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]
# the 69 other sequences of hg19.fa: chrM, the 9 alternate haplotypes, and 59 unplaced / unlocalized contigs.
# The haplotype and chrM lengths are hg19's; the 59 contig lengths are drawn (fixed seed) from hg19's range
# of 15-211 kb so that the assembly totals hg19's 3,137,161,264 bp.
HG19_EXTRA = [("chrM", 16571), ("chr6_apd_hap1", 4622290), ("chr6_cox_hap2", 4795371), ("chr6_dbb_hap3", 4610396),
              ("chr6_mann_hap4", 4683263), ("chr6_mcf_hap5", 4833398), ("chr6_qbl_hap6", 4611984),
              ("chr6_ssto_hap7", 4928567), ("chr4_ctg9_hap1", 590426), ("chr17_ctg5_hap1", 1680828)]
HG19_TOTAL = 3137161264


def hg19_sequences():
    """(names, lengths) of the 93 sequences."""
    import random
    rnd = random.Random(19)
    rest = HG19_TOTAL - sum(HG19_CHROMS) - sum(l for _, l in HG19_EXTRA)
    n = 59
    w = [rnd.uniform(15000, 211000) for _ in range(n)]
    s = sum(w)
    gl = [max(15008, int(x * rest / s)) for x in w]
    gl[-1] += rest - sum(gl)
    names = HG19_CHROM_NAMES + [nm for nm, _ in HG19_EXTRA] + ["chrUn_gl%06d" % (191 + i) for i in range(n)]
    lens = HG19_CHROMS + [l for _, l in HG19_EXTRA] + gl
    assert sum(lens) == HG19_TOTAL and len(lens) == 93
    return names, lens


def _rand_codes(torch, g, dev, L):
    codes = torch.empty(L, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for s in range(0, L, step):
        e = min(L, s + step)
        codes[s:e] = torch.randint(0, 4, (e - s,), generator=g, device=dev, dtype=torch.uint8)
    return codes


def _to_ascii(torch, dev, codes):
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    L = codes.numel()
    step = 1 << 28
    if L <= step:
        return lut[codes.long()]
    out = torch.empty(L, dtype=torch.uint8, device=dev)
    for s in range(0, L, step):
        out[s:s + step] = lut[codes[s:s + step].long()]
    return out


def _scatter(torch, codes, idx, vals):
    """codes[idx] = vals with a defined outcome where idx repeats (the last one wins): overlapping copies
    would otherwise make the genome differ from run to run and from rank to rank."""
    si, perm = torch.sort(idx.reshape(-1), stable=True)
    sv = vals.reshape(-1)[perm]
    keep = torch.ones_like(si, dtype=torch.bool)
    keep[:-1] = si[1:] != si[:-1]
    codes[si[keep]] = sv[keep]


def _mutate(torch, g, dev, vals, div):
    """vals [m, l] codes, div [m] per-copy substitution probability."""
    mut = torch.rand(vals.shape, generator=g, device=dev) < div[:, None]
    rnd = torch.randint(1, 4, vals.shape, generator=g, device=dev, dtype=torch.uint8)
    return torch.where(mut, (vals + rnd) & 3, vals)


def _plant(torch, g, dev, codes, src, src_off, lens, div, starts, indel_per_sub=0.0, budget=1 << 25):
    """Write mutated copies into `codes`: copy i = src[src_off[i] : src_off[i] + lens[i]] with per-base substitution
    probability div[i] and about div * len * indel_per_sub single-base indels, reverse-complemented for
    every other copy, at codes[starts[i] ...].  Copies are processed longest first in padded chunks."""
    n = lens.numel()
    if n == 0:
        return
    L = codes.numel()
    order = torch.argsort(lens, descending=True)
    lens, div, starts, src_off = lens[order], div[order], starts[order], src_off[order]
    rc_all = (torch.arange(n, device=dev) & 1).bool()
    lens_host = lens.cpu()
    i = 0
    while i < n:
        lmax = int(lens_host[i])
        m = max(1, min(n - i, budget // max(1, lmax)))
        ln, dd, st, so, rc = lens[i:i + m], div[i:i + m], starts[i:i + m], src_off[i:i + m], rc_all[i:i + m]
        ar = torch.arange(lmax, device=dev)
        sidx = so[:, None] + ar[None, :]
        if indel_per_sub > 0:
            shift = torch.zeros((m, lmax), dtype=torch.int16, device=dev)
            n_ev = torch.poisson((dd * ln.float() * indel_per_sub).clamp(max=8.0), generator=g).long().clamp(max=6)
            rows = torch.arange(m, device=dev)
            for e in range(6):
                has = n_ev > e
                if not bool(has.any()):
                    break
                p = (torch.rand(m, generator=g, device=dev) * ln.float()).long().clamp(max=lmax - 1)
                sign = (torch.randint(0, 2, (m,), generator=g, device=dev) * 2 - 1).to(torch.int16)
                shift[rows[has], p[has]] += sign[has]
            sidx = sidx + torch.cumsum(shift, 1).long()
        vals = src[sidx.clamp(0, src.numel() - 1)]
        vals = _mutate(torch, g, dev, vals, dd)
        # reverse complement for odd copies: placed[j] = 3 - vals[len-1-j]
        jj = torch.where(rc[:, None], ln[:, None] - 1 - ar[None, :], ar[None, :]).clamp(min=0)
        vals = torch.where(rc[:, None], 3 - vals.gather(1, jj), vals)
        valid = ar[None, :] < ln[:, None]
        idx = (st[:, None] + ar[None, :])
        valid &= idx < L
        _scatter(torch, codes, idx[valid], vals[valid])
        i += m


def _uniform(torch, g, dev, n, lo, hi):
    return lo + (hi - lo) * torch.rand(n, generator=g, device=dev)


def _age_mix(torch, g, dev, n, classes):
    """classes: [(share, d_lo, d_hi)] -> per-copy divergence from the consensus."""
    u = torch.rand(n, generator=g, device=dev)
    d = torch.zeros(n, device=dev)
    acc = 0.0
    for share, lo, hi in classes:
        sel = (u >= acc) & (u < acc + share)
        d = torch.where(sel, _uniform(torch, g, dev, n, lo, hi), d)
        acc += share
    return d


def genome_hg19like(torch, dev, scale, seed, contigs=0, single=None):
    names, lens = hg19_sequences()
    if scale != 1.0:
        lens = [max(600, int(l * scale)) for l in lens]
    if contigs:
        total = sum(lens)
        lens = [total // contigs] * (contigs - 1) + [total - (total // contigs) * (contigs - 1)]
        names = ["ctg%d" % i for i in range(contigs)]
    if single:  # one sequence of exactly this many bases, the families scaled to it (BASELINE configs[0]: a chr2-length genome)
        names, lens = [single[0]], [int(single[1])]
    L = sum(lens)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    codes = _rand_codes(torch, g, dev, L)
    f = L / float(HG19_TOTAL)

    def starts_for(n, span):
        return torch.randint(0, max(1, L - span - 64), (n,), generator=g, device=dev)

    # --- SINE-like family: 1.1 M x 300 bp
    n_alu = int(1_100_000 * f)
    if n_alu:
        cons = torch.randint(0, 4, (300,), generator=g, device=dev, dtype=torch.uint8)
        d = _age_mix(torch, g, dev, n_alu, [(0.08, 0.02, 0.06), (0.57, 0.08, 0.14), (0.35, 0.14, 0.20)])
        ln = torch.full((n_alu,), 300, dtype=torch.long, device=dev)
        trunc = torch.rand(n_alu, generator=g, device=dev) < 0.3  # a third of the copies are 5'-truncated
        ln = torch.where(trunc, (100 + 200 * torch.rand(n_alu, generator=g, device=dev)).long(), ln)
        _plant(torch, g, dev, codes, cons, 300 - ln, ln, d, starts_for(n_alu, 300), indel_per_sub=0.12)
    # --- LINE-like family: 0.5 M copies, 3' ends of a 6 kb consensus, mean ~1 kb
    n_l1 = int(500_000 * f)
    if n_l1:
        cons = torch.randint(0, 4, (6000,), generator=g, device=dev, dtype=torch.uint8)
        ln = (300 + torch.empty(n_l1, device=dev).exponential_(1.0 / 750.0, generator=g)).long().clamp(max=6000)
        d = _age_mix(torch, g, dev, n_l1, [(0.10, 0.01, 0.05), (0.90, 0.06, 0.20)])
        _plant(torch, g, dev, codes, cons, 6000 - ln, ln, d, starts_for(n_l1, 6000), indel_per_sub=0.12)
    # --- satellites: tandem arrays of higher-order units built from a 171 bp monomer
    sat_bp = int(0.025 * L)
    mean_arr = min(275_000, max(2_000, L // 400))
    n_arr = max(1, sat_bp // mean_arr)
    mono = torch.randint(0, 4, (171,), generator=g, device=dev, dtype=torch.uint8)
    shared_units = []
    for k in range(3):  # three families whose arrays share one unit each
        nm = 4 + 4 * k
        u = mono.repeat(nm)
        u = _mutate(torch, g, dev, u[None, :], torch.tensor([0.2], device=dev))[0]
        shared_units.append(u)
    arr_len = _uniform(torch, g, dev, n_arr, 0.18 * mean_arr, 1.82 * mean_arr).long()
    arr_start = starts_for(n_arr, int(1.82 * mean_arr))
    arr_len_h, arr_start_h = arr_len.cpu().tolist(), arr_start.cpu().tolist()
    for a in range(n_arr):
        if a % 2 == 0:  # private unit: this array only
            nm = 2 + (a * 7) % 11
            u = _mutate(torch, g, dev, mono.repeat(nm)[None, :], torch.tensor([0.25], device=dev))[0]
        else:
            u = shared_units[(a // 2) % 3]
        ul = u.numel()
        copies = max(1, arr_len_h[a] // ul)
        dd = torch.full((copies,), 0.01 + 0.02 * ((a * 13) % 10) / 10.0, device=dev)
        st = arr_start_h[a] + torch.arange(copies, device=dev) * ul
        vals = _mutate(torch, g, dev, u[None, :].expand(copies, ul), dd)
        idx = st[:, None] + torch.arange(ul, device=dev)[None, :]
        ok = idx < L
        _scatter(torch, codes, idx[ok], vals[ok])
    # --- simple repeats: 0.5 M loci of 1-6 bp motifs + homopurine / homopyrimidine tracts
    n_ssr = int(500_000 * f)
    if n_ssr:
        ml = torch.randint(1, 7, (n_ssr,), generator=g, device=dev)
        motif = torch.randint(0, 4, (n_ssr, 6), generator=g, device=dev, dtype=torch.uint8)
        ln = torch.randint(20, 121, (n_ssr,), generator=g, device=dev)
        st = starts_for(n_ssr, 128)
        for c0 in range(0, n_ssr, 1 << 18):
            sl = slice(c0, c0 + (1 << 18))
            ar = torch.arange(120, device=dev)
            vals = motif[sl].gather(1, (ar[None, :] % ml[sl, None]))
            vals = _mutate(torch, g, dev, vals, torch.full((vals.shape[0],), 0.03, device=dev))
            valid = ar[None, :] < ln[sl, None]
            idx = st[sl, None] + ar[None, :]
            _scatter(torch, codes, idx[valid], vals[valid])
    n_tract = int(250_000 * f)
    if n_tract:  # pyrimidine (C/T) or purine (A/G) tracts: ONE letter in the C->T resp. G->A converted strand
        ln = torch.randint(40, 161, (n_tract,), generator=g, device=dev)
        st = starts_for(n_tract, 192)
        pur = torch.rand(n_tract, generator=g, device=dev) < 0.5
        for c0 in range(0, n_tract, 1 << 18):
            sl = slice(c0, c0 + (1 << 18))
            ar = torch.arange(160, device=dev)
            m = ln[sl].numel()
            bit = torch.randint(0, 2, (m, 160), generator=g, device=dev, dtype=torch.uint8)
            vals = torch.where(pur[sl, None], bit * 2, 1 + bit * 2)  # A/G = 0/2, C/T = 1/3
            valid = ar[None, :] < ln[sl, None]
            idx = st[sl, None] + ar[None, :]
            _scatter(torch, codes, idx[valid], vals[valid])
    # --- segmental duplications: 5 % of the genome, 5-50 kb blocks (capped for small test genomes) at 0-2 %
    blk_hi = int(min(50_000, max(600, L // 200)))
    blk_lo = max(300, blk_hi // 10)
    n_sd = int(0.05 * L / (0.5 * (blk_lo + blk_hi)))
    if n_sd:
        ln = torch.randint(blk_lo, blk_hi + 1, (n_sd,), generator=g, device=dev)
        src_off = starts_for(n_sd, blk_hi)
        dst = starts_for(n_sd, blk_hi)
        d = _uniform(torch, g, dev, n_sd, 0.0, 0.02)
        snap = codes.clone()  # copies are taken from the genome as it is now
        _plant(torch, g, dev, codes, snap, src_off, ln, d, dst, indel_per_sub=0.05)
        del snap
    return _to_ascii(torch, dev, codes), lens, names


def genome_easy(torch, dev, scale, seed, contigs=0):
    """Round-1 benchmark genome: iid bases + four small repeat families (kept for continuity)."""
    lens = [max(1000, int(l * scale)) for l in HG19_CHROMS]
    names = list(HG19_CHROM_NAMES)
    if contigs:
        total = sum(lens)
        lens = [total // contigs] * (contigs - 1) + [total - (total // contigs) * (contigs - 1)]
        names = ["ctg%d" % i for i in range(contigs)]
    L = sum(lens)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    codes = _rand_codes(torch, g, dev, L)

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
    return _to_ascii(torch, dev, codes), lens, names


CHR2_LEN = HG19_CHROMS[1]  # 243,199,373: SURVEY 8(d) config 1


def make_genome(torch, dev, scale, seed, kind="hg19like", contigs=0):
    """-> (ASCII genome tensor on dev, sequence lengths, sequence names)"""
    if kind == "chr2like":
        return genome_hg19like(torch, dev, 1.0, seed, 0, single=("chr2", CHR2_LEN))
    if kind == "easy":
        return genome_easy(torch, dev, scale, seed, contigs)
    if kind != "hg19like":
        raise ValueError("genome kind must be hg19like or easy")
    return genome_hg19like(torch, dev, scale, seed, contigs)


def _comp_table(torch, dev):
    comp = torch.zeros(256, dtype=torch.uint8, device=dev)
    for a, b in ((65, 84), (67, 71), (71, 67), (84, 65)):
        comp[a] = b
    return comp


LOWQ_SHARE, LOWQ_LO, LOWQ_HI = 0.06, 0.04, 0.14  # quality tail of a real run: 6 % of the reads at 4-14 % errors


def _error_rate(torch, g, dev, m, lowq):
    """per-read substitution probability: 1 %, or the low-quality tail (real data sets leave ~10 % of the
    reads unmapped, reference doc/Supplementary Data.pdf Table S4; clean synthetic reads would all map)"""
    e = torch.full((m,), 0.01, device=dev)
    if lowq:
        tail = torch.rand(m, generator=g, device=dev) < LOWQ_SHARE
        e = torch.where(tail, LOWQ_LO + (LOWQ_HI - LOWQ_LO) * torch.rand(m, generator=g, device=dev), e)
    return e


def make_reads(torch, dev, genome_ascii, n, read_len, seed, ag=False, lowq=True):
    """n x read_len ASCII reads on the device: uniform positions, both strands, 95 % C->T (ag: G->A, the A-rich
    strand of -A / PBAT libraries), 1 % substitutions (lowq: 6 % of the reads 4-14 %)."""
    L = genome_ascii.numel()
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = torch.empty((n, read_len), dtype=torch.uint8, device=dev)
    comp = _comp_table(torch, dev)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    ar = torch.arange(read_len, device=dev)
    src, dst = (71, 65) if ag else (67, 84)
    chunk = 1 << 22
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        pos = torch.randint(0, L - read_len, (m,), generator=g, device=dev)
        r = genome_ascii[pos[:, None] + ar[None, :]]
        rev = torch.rand(m, generator=g, device=dev) < 0.5
        rc = comp[r.flip(1).long()]
        r = torch.where(rev[:, None], rc, r)
        conv = (r == src) & (torch.rand(r.shape, generator=g, device=dev) < 0.95)
        r = torch.where(conv, torch.full_like(r, dst), r)
        sub = torch.rand(r.shape, generator=g, device=dev) < _error_rate(torch, g, dev, m, lowq)[:, None]
        rnd = lut[torch.randint(0, 4, r.shape, generator=g, device=dev)]
        r = torch.where(sub, rnd, r)
        out[s:s + m] = r
    offsets = torch.arange(n + 1, device=dev, dtype=torch.int64) * read_len
    return out.reshape(-1), offsets


def make_pairs(torch, dev, genome_ascii, n, read_len, seed, frag_lo=120, frag_hi=500, lowq=True):
    """n pairs on the device: fragment length U[max(frag_lo, read_len), frag_hi] from either strand, bisulfite
    (95 % C->T on the fragment), mate 1 = fragment[:L], mate 2 = revcomp(fragment)[:L], 1 % substitutions
    (lowq: 6 % of the mates 4-14 %, drawn per mate)."""
    L = genome_ascii.numel()
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    m1 = torch.empty((n, read_len), dtype=torch.uint8, device=dev)
    m2 = torch.empty((n, read_len), dtype=torch.uint8, device=dev)
    comp = _comp_table(torch, dev)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)
    ar = torch.arange(read_len, device=dev)
    chunk = 1 << 21
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        flen = torch.randint(max(frag_lo, read_len), frag_hi + 1, (m,), generator=g, device=dev)
        pos = torch.randint(0, L - frag_hi - 100, (m,), generator=g, device=dev)
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
        for r, dstt in ((r1, m1), (r2, m2)):
            sub = torch.rand(r.shape, generator=g, device=dev) < _error_rate(torch, g, dev, m, lowq)[:, None]
            rnd = lut[torch.randint(0, 4, r.shape, generator=g, device=dev)]
            dstt[s:s + m] = torch.where(sub, rnd, r)
    offsets = torch.arange(n + 1, device=dev, dtype=torch.int64) * read_len
    return m1.reshape(-1), m2.reshape(-1), offsets
