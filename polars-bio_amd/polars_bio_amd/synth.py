"""Synthetic interval sets of the BASELINE.json shapes.

numpy ``Generator(PCG64(seed))``, seed 42 for the probe side and 43 for the build side;
24 contigs chr1..chr22,chrX,chrY with GRCh38 lengths; rows assigned to contigs in
proportion to length; ``start ~ U[0, len_c - L)``; probe length ``L ~ U{100..150}``
("short reads"), build length ``L ~ U{200..2000}`` (exon-like); int32, 0-based half-open
(FilterOp.Strict); rows are NOT sorted (contigs and positions are shuffled).
"""
from __future__ import annotations

import numpy as np

GRCH38 = {
    "chr1": 248956422, "chr2": 242193529, "chr3": 198295559, "chr4": 190214555, "chr5": 181538259,
    "chr6": 170805979, "chr7": 159345973, "chr8": 145138636, "chr9": 138394717, "chr10": 133797422,
    "chr11": 135086622, "chr12": 133275309, "chr13": 114364328, "chr14": 107043718, "chr15": 101991189,
    "chr16": 90338345, "chr17": 83257441, "chr18": 80373285, "chr19": 58617616, "chr20": 64444167,
    "chr21": 46709983, "chr22": 50818468, "chrX": 156040895, "chrY": 57227415,
}
CONTIG_NAMES = list(GRCH38)
CONTIG_LENGTHS = np.array([GRCH38[c] for c in CONTIG_NAMES], dtype=np.int64)

PROBE_LEN = (100, 150)
BUILD_LEN = (200, 2000)
DENSE_BUILD_LEN = (5000, 40000)


def make_side(n: int, seed: int, len_range, n_contigs: int = 24):
    """-> (contig_id int32[n], start int32[n], end int32[n]), unsorted."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lengths = CONTIG_LENGTHS[:n_contigs]
    p = lengths / lengths.sum()
    # contig of every row: multinomial split, then shuffled so contigs interleave
    contig = rng.choice(n_contigs, size=n, p=p).astype(np.int32) if n_contigs > 1 else np.zeros(n, np.int32)
    L = rng.integers(len_range[0], len_range[1] + 1, size=n, dtype=np.int64)
    u = rng.random(n)
    start = np.floor(u * (lengths[contig] - L)).astype(np.int64)
    end = start + L
    return contig, start.astype(np.int32), end.astype(np.int32)


WORKLOADS = {
    "overlap_1k_1k_1contig": (1_000, 1_000, 1, BUILD_LEN),
    "overlap_10M_1M_1contig": (10_000_000, 1_000_000, 1, BUILD_LEN),
    "overlap_100M_5M_24contig": (100_000_000, 5_000_000, 24, BUILD_LEN),
    "nearest_50M_2M_24contig": (50_000_000, 2_000_000, 24, BUILD_LEN),
    "count_200M_200k_24contig": (200_000_000, 200_000, 24, BUILD_LEN),
    "overlap_100M_5M_24contig_dense": (100_000_000, 5_000_000, 24, DENSE_BUILD_LEN),
}


def workload(name: str, scale: float = 1.0):
    """BASELINE.json configs -> (probe, build, n_contigs): the ONE input every rank count works on (make_rows below: a rank of an
    N > 1 run holds the rows of its contigs out of exactly these columns, global row = position here)."""
    np_, nb, nc, blen = WORKLOADS[name]
    np_, nb = max(1, int(np_ * scale)), max(1, int(nb * scale))
    return make_rows(np_, 42, PROBE_LEN, nc)[0], make_rows(nb, 43, blen, nc)[0], nc


def expected_pairs(n_probe: int, n_build: int, n_contigs: int = 24, build_len=BUILD_LEN) -> float:
    """E[pairs] ~= sum_c Np_c * Nb_c * (E[Lp] + E[Lb]) / len_c (Strict)."""
    lengths = CONTIG_LENGTHS[:n_contigs].astype(np.float64)
    p = lengths / lengths.sum()
    el = (PROBE_LEN[0] + PROBE_LEN[1]) / 2 + (build_len[0] + build_len[1]) / 2
    return float(((n_probe * p) * (n_build * p) * el / lengths).sum())


# ---- the same distribution, generated shard by shard (bench.py at N > 1) ---------------------------------------------
# make_side draws the whole side from ONE generator, so a rank that wants its contigs only would still have to draw all n
# rows (100 M rows x N ranks of host work before the first kernel).  The sharded generator fixes the rows per contig
# (largest remainder of n * len_c / sum len) and draws every contig in blocks of SHARD_BLOCK rows from its own stream
# SeedSequence([seed, contig, block]): any rank can produce any row range of any contig, every rank agrees on the data, and
# nothing outside the shard is ever drawn.  Global row id of row i of contig c = contig_offsets[c] + i.
SHARD_BLOCK = 1 << 20


def contig_rows(n: int, n_contigs: int = 24) -> np.ndarray:
    """Rows per contig, proportional to the contig lengths (largest remainder), summing to n."""
    lengths = CONTIG_LENGTHS[:n_contigs].astype(np.float64)
    share = n * lengths / lengths.sum()
    rows = np.floor(share).astype(np.int64)
    rest = int(n - rows.sum())
    if rest:
        rows[np.argsort(-(share - rows), kind="stable")[:rest]] += 1
    return rows


def make_contig_rows(c: int, lo: int, hi: int, seed: int, len_range, n_contigs: int = 24):
    """Rows lo .. hi-1 of contig c -> (start int32, end int32)."""
    length = int(CONTIG_LENGTHS[c])
    s_parts, e_parts = [], []
    for k in range(lo // SHARD_BLOCK, (max(hi, lo + 1) - 1) // SHARD_BLOCK + 1):
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, c, k])))
        L = rng.integers(len_range[0], len_range[1] + 1, size=SHARD_BLOCK, dtype=np.int64)
        start = np.floor(rng.random(SHARD_BLOCK) * (length - L)).astype(np.int64)
        a, b = max(lo, k * SHARD_BLOCK) - k * SHARD_BLOCK, min(hi, (k + 1) * SHARD_BLOCK) - k * SHARD_BLOCK
        if b > a:
            s_parts.append(start[a:b]); e_parts.append((start + L)[a:b])
    if not s_parts:
        return np.empty(0, np.int32), np.empty(0, np.int32)
    return np.concatenate(s_parts).astype(np.int32), np.concatenate(e_parts).astype(np.int32)


def make_shard(n: int, seed: int, len_range, n_contigs: int, pieces, shuffle_seed=None):
    """pieces: [(contig, lo, hi)] row ranges -> ((contig, start, end) int32 arrays, global row ids int32).
    shuffle_seed: permute the shard's rows (the N = 1 workload is unsorted with interleaved contigs; so is every shard)."""
    rows = contig_rows(n, n_contigs)
    offs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    cs, ss, es, ids = [], [], [], []
    for c, lo, hi in pieces:
        hi = min(hi, int(rows[c]))
        if hi <= lo:
            continue
        s, e = make_contig_rows(c, lo, hi, seed, len_range, n_contigs)
        cs.append(np.full(hi - lo, c, np.int32)); ss.append(s); es.append(e)
        ids.append((offs[c] + np.arange(lo, hi, dtype=np.int64)).astype(np.int32))
    if not cs:
        z = np.empty(0, np.int32)
        return (z, z, z), z
    c, s, e, i = np.concatenate(cs), np.concatenate(ss), np.concatenate(es), np.concatenate(ids)
    if shuffle_seed is not None:
        p = np.random.Generator(np.random.PCG64(shuffle_seed)).permutation(len(c))
        c, s, e, i = c[p], s[p], e[p], i[p]
    return (c, s, e), i


# ---- ONE input for every rank count (round 6) ------------------------------------------------------------------------------
# make_side draws a side from one stream, so a rank that wants its contigs only has to draw all n rows; make_shard above draws contig by
# contig but numbers the rows contig after contig -- a different table from the N = 1 one.  make_rows defines the side ONCE, in a form
# any rank can cut its share out of without drawing the rest:
#   * the contig of global row g is a pure function of (seed, g): a 64-bit mix of g (splitmix64 finaliser) against the cumulative contig
#     shares -- contigs interleave like make_side's multinomial draw, nothing is sorted;
#   * the (start, end) of the j-th row of contig c come from that contig's own block streams (make_contig_rows, as make_shard).
# A shard = the rows whose contig is in `contigs` (and whose global row lies in `row_range`), IN GLOBAL ROW ORDER, with their global row
# ids -- what ivj_host_shard makes of the full table; `workload()` is the same call without a selection.  The N = 1 bench, the N > 1 bench
# and the 8-rank dry run therefore join one and the same table (tests/test_synth.py: every partition of the contigs reassembles to it).
_MIX, _WEYL = np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x9E3779B97F4A7C15)
_ROW_CHUNK = 1 << 20


def _threads() -> int:
    import os
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        lim = os.cpu_count() if q == "max" else max(1, int(float(q) / float(p)))
    except Exception:
        lim = os.cpu_count() or 1
    return max(1, min(16, lim, os.cpu_count() or 1))


def _pool_map(fn, items):
    from concurrent.futures import ThreadPoolExecutor
    items = list(items)
    if len(items) <= 1 or _threads() == 1:
        return [fn(x) for x in items]
    with ThreadPoolExecutor(_threads()) as ex:                   # numpy releases the GIL inside its loops
        return list(ex.map(fn, items))


def row_contigs(n: int, seed: int, n_contigs: int = 24) -> np.ndarray:
    """Contig id (uint8) of every global row 0 .. n-1 of the side `seed`: the top 32 bits of a multiply-xorshift-multiply mix of
    (row + seed) against the cumulative contig shares."""
    out = np.zeros(n, np.uint8)
    if n_contigs <= 1 or n == 0:
        return out
    lengths = CONTIG_LENGTHS[:n_contigs].astype(np.float64)
    thr = np.minimum(np.floor(np.cumsum(lengths / lengths.sum())[:-1] * 2.0 ** 32), 2.0 ** 32 - 1).astype(np.uint32)   # upper edges of contigs 0 .. nc-2

    def part(lo):
        hi = min(n, lo + _ROW_CHUNK)
        with np.errstate(over="ignore"):
            z = np.arange(lo, hi, dtype=np.uint64)
            z += np.uint64(seed)
            z *= _WEYL
            z ^= z >> np.uint64(32)
            z *= _MIX
            z >>= np.uint64(32)
        out[lo:hi] = np.searchsorted(thr, z.astype(np.uint32), side="right")

    _pool_map(part, range(0, n, _ROW_CHUNK))
    return out


def contig_counts(n: int, seed: int, n_contigs: int = 24) -> np.ndarray:
    """Rows per contig of the side `seed` (the LPT weights of the contig sharding)."""
    return np.bincount(row_contigs(n, seed, n_contigs), minlength=n_contigs).astype(np.int64)


def make_rows(n: int, seed: int, len_range, n_contigs: int = 24, contigs=None, row_range=None):
    """-> ((contig, start, end) int32 arrays, global row ids int32 ascending) of the rows of the side `seed` whose contig is in
    `contigs` (None: all) and whose global row lies in row_range = (lo, hi) (None: all)."""
    cg = row_contigs(n, seed, n_contigs)
    lo_g, hi_g = (0, n) if row_range is None else (max(0, int(row_range[0])), min(n, int(row_range[1])))
    wanted = list(range(n_contigs)) if contigs is None else sorted(set(int(x) for x in contigs))

    def one(c):
        pos = np.flatnonzero(cg == np.uint8(c))                 # the contig's rows, ascending
        k_lo, k_hi = (0, len(pos)) if row_range is None else (int(np.searchsorted(pos, lo_g)), int(np.searchsorted(pos, hi_g)))
        if k_hi <= k_lo:
            return None
        s, e = make_contig_rows(c, k_lo, k_hi, seed, len_range, n_contigs)
        return c, pos[k_lo:k_hi], s, e

    parts = [p for p in _pool_map(one, wanted) if p is not None]
    if contigs is None and row_range is None:
        start, end = np.empty(n, np.int32), np.empty(n, np.int32)
        for _, pos, s, e in parts:
            start[pos] = s; end[pos] = e
        return (cg.astype(np.int32), start, end), np.arange(n, dtype=np.int32)
    if not parts:
        z = np.empty(0, np.int32)
        return (z, z, z), z
    g = np.concatenate([p[1] for p in parts])
    o = np.argsort(g, kind="stable")                            # back to global row order
    cs = np.concatenate([np.full(len(p[1]), p[0], np.int32) for p in parts])
    return (cs[o], np.concatenate([p[2] for p in parts])[o], np.concatenate([p[3] for p in parts])[o]), g[o].astype(np.int32)
