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


def workload(name: str):
    """BASELINE.json configs -> (probe, build, n_contigs)."""
    cfg = {
        "overlap_1k_1k_1contig": (1_000, 1_000, 1, BUILD_LEN),
        "overlap_10M_1M_1contig": (10_000_000, 1_000_000, 1, BUILD_LEN),
        "overlap_100M_5M_24contig": (100_000_000, 5_000_000, 24, BUILD_LEN),
        "nearest_50M_2M_24contig": (50_000_000, 2_000_000, 24, BUILD_LEN),
        "count_200M_200k_24contig": (200_000_000, 200_000, 24, BUILD_LEN),
        "overlap_100M_5M_24contig_dense": (100_000_000, 5_000_000, 24, DENSE_BUILD_LEN),
    }[name]
    np_, nb, nc, blen = cfg
    return make_side(np_, 42, PROBE_LEN, nc), make_side(nb, 43, blen, nc), nc


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
