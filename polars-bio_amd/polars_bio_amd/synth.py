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
