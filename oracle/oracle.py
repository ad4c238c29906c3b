"""CPU oracle for the interval-join hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / the timed baseline.
The product path (``polars_bio_amd``) never imports it and fails loudly when
the HIP library is missing.

Two independent restatements live here:

* ``libivj_oracle.so`` (``ivj_oracle.c``): brute-force O(Np*Nb) definitions and
  a sort + bound-search implementation (also the timed CPU baseline).
* numpy restatements (``np_*``) of the same semantics written directly from
  the reference's Python-visible behaviour:
    - Strict/Weak ........ /root/reference/polars_bio/range_op.py:56-84
    - two-rank count ..... /root/reference/polars_bio/range_op.py:548-595
    - nearest distance ... /root/reference/tests/_expected.py:130-172

Pinned against the reference's golden tables by ``tests/test_oracle_golden.py``
(SURVEY.md section 8c).  Unpinned in the reference (stated in DESIGN.md):
nearest k>1, nearest ties at equal non-zero distance, overlap=False, rows whose
contig is absent from the other side, output row order.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Side(C.Structure):
    _fields_ = [
        ("contig", C.c_void_p),
        ("start", C.c_void_p),
        ("end", C.c_void_p),
        ("n", C.c_int64),
    ]


def build_lib(force: bool = False) -> str:
    """Compile ivj_oracle.c with gcc (idempotent)."""
    so = os.path.join(_HERE, "libivj_oracle.so")
    src = os.path.join(_HERE, "ivj_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libivj_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libivj_oracle.so")
        if not os.path.exists(so):
            build_lib()
        L = C.CDLL(so)
        P = C.POINTER(_Side)
        L.orc_overlap_brute.restype = C.c_int64
        L.orc_overlap_brute.argtypes = [P, P, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_count_overlaps_brute.restype = None
        L.orc_count_overlaps_brute.argtypes = [P, P, C.c_int, C.c_void_p]
        L.orc_nearest_brute.restype = None
        L.orc_nearest_brute.argtypes = [P, P, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_index_build.restype = C.c_void_p
        L.orc_index_build.argtypes = [P, C.c_int]
        L.orc_index_free.restype = None
        L.orc_index_free.argtypes = [C.c_void_p]
        L.orc_count_overlaps_fast.restype = None
        L.orc_count_overlaps_fast.argtypes = [C.c_void_p, P, C.c_int, C.c_void_p, C.c_int]
        L.orc_overlap_fast.restype = C.c_int64
        L.orc_overlap_fast.argtypes = [C.c_void_p, P, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.orc_nearest_fast.restype = None
        L.orc_nearest_fast.argtypes = [C.c_void_p, P, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


class Side:
    """(contig id, start, end) int32 columns kept alive for the ctypes call."""

    def __init__(self, contig, start, end):
        self.contig, self.start, self.end = _i32(contig), _i32(start), _i32(end)
        assert self.contig.shape == self.start.shape == self.end.shape
        self.n = int(self.contig.shape[0])
        self.c = _Side(self.contig.ctypes.data, self.start.ctypes.data, self.end.ctypes.data, self.n)

    def ref(self):
        return C.byref(self.c)


# --------------------------------------------------------------------------
# C oracle wrappers
# --------------------------------------------------------------------------

def overlap_brute(probe: Side, build: Side, strict: bool):
    L = lib()
    n = L.orc_overlap_brute(probe.ref(), build.ref(), int(strict), None, None, 0)
    p = np.empty(n, np.int32)
    b = np.empty(n, np.int32)
    L.orc_overlap_brute(probe.ref(), build.ref(), int(strict), p.ctypes.data, b.ctypes.data, n)
    return p, b


def count_overlaps_brute(probe: Side, build: Side, strict: bool):
    out = np.empty(probe.n, np.int64)
    lib().orc_count_overlaps_brute(probe.ref(), build.ref(), int(strict), out.ctypes.data)
    return out


def nearest_brute(probe: Side, build: Side, strict: bool, k: int = 1, include_overlaps: bool = True):
    idx = np.empty((probe.n, k), np.int32)
    dist = np.empty((probe.n, k), np.int64)
    n = np.empty(probe.n, np.int32)
    lib().orc_nearest_brute(probe.ref(), build.ref(), int(strict), k, int(include_overlaps),
                            idx.ctypes.data, dist.ctypes.data, n.ctypes.data)
    return idx, dist, n


class Index:
    """Sorted build side (sort + bound search path)."""

    def __init__(self, build: Side, n_contigs: int):
        self.build = build
        self.h = lib().orc_index_build(build.ref(), int(n_contigs))

    def close(self):
        if self.h:
            lib().orc_index_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def count_overlaps_fast(ix: Index, probe: Side, strict: bool, threads: int = 0):
    out = np.empty(probe.n, np.int64)
    lib().orc_count_overlaps_fast(ix.h, probe.ref(), int(strict), out.ctypes.data, threads)
    return out


def overlap_fast(ix: Index, probe: Side, strict: bool, threads: int = 0, count_only: bool = False):
    L = lib()
    n = L.orc_overlap_fast(ix.h, probe.ref(), int(strict), None, None, 0, threads)
    if count_only:
        return n
    p = np.empty(n, np.int32)
    b = np.empty(n, np.int32)
    L.orc_overlap_fast(ix.h, probe.ref(), int(strict), p.ctypes.data, b.ctypes.data, n, threads)
    return p, b


def nearest_fast(ix: Index, probe: Side, strict: bool, k: int = 1, include_overlaps: bool = True,
                 threads: int = 0):
    idx = np.empty((probe.n, k), np.int32)
    dist = np.empty((probe.n, k), np.int64)
    n = np.empty(probe.n, np.int32)
    lib().orc_nearest_fast(ix.h, probe.ref(), int(strict), k, int(include_overlaps),
                           idx.ctypes.data, dist.ctypes.data, n.ctypes.data, threads)
    return idx, dist, n


# --------------------------------------------------------------------------
# numpy restatements (independent of the C code)
# --------------------------------------------------------------------------

def np_count_overlaps(probe: Side, build: Side, strict: bool) -> np.ndarray:
    """Two-rank formula of the SQL sweep (range_op.py:548-595):
    Strict: #{s2 < e1} - #{e2 <= s1};  Weak: #{s2 <= e1} - #{e2 < s1}.
    Valid for non-inverted rows that are not (Strict) zero-length on both sides."""
    out = np.zeros(probe.n, np.int64)
    for c in np.unique(probe.contig):
        pm = probe.contig == c
        bm = build.contig == c
        if not bm.any():
            continue
        s2 = np.sort(build.start[bm])
        e2 = np.sort(build.end[bm])
        if strict:
            out[pm] = np.searchsorted(s2, probe.end[pm], "left") - np.searchsorted(e2, probe.start[pm], "right")
        else:
            out[pm] = np.searchsorted(s2, probe.end[pm], "right") - np.searchsorted(e2, probe.start[pm], "left")
    return out


def np_overlap_pairs(probe: Side, build: Side, strict: bool):
    """Dense boolean-matrix definition; small inputs only."""
    same = probe.contig[:, None] == build.contig[None, :]
    if strict:
        m = same & (probe.start[:, None] < build.end[None, :]) & (build.start[None, :] < probe.end[:, None])
    else:
        m = same & (probe.start[:, None] <= build.end[None, :]) & (build.start[None, :] <= probe.end[:, None])
    p, b = np.nonzero(m)
    order = np.lexsort((b, build.start[b], p))
    return p[order].astype(np.int32), b[order].astype(np.int32)


def np_nearest_distance(probe: Side, build: Side, strict: bool) -> np.ndarray:
    """k=1 distance only (what tests/test_bioframe.py:168-186 compares):
    0 when any row overlaps, else min over rows of max(b.start-q.end, q.start-b.end);
    -1 when the contig is absent from the build side."""
    out = np.full(probe.n, -1, np.int64)
    for i in range(probe.n):
        bm = build.contig == probe.contig[i]
        if not bm.any():
            continue
        bs = build.start[bm].astype(np.int64)
        be = build.end[bm].astype(np.int64)
        qs, qe = int(probe.start[i]), int(probe.end[i])
        ov = ((qs < be) & (bs < qe)) if strict else ((qs <= be) & (bs <= qe))
        if ov.any():
            out[i] = 0
        else:
            out[i] = np.maximum(bs - qe, qs - be).min()
    return out


def encode_contigs(*cols):
    """Shared dictionary encoding of chrom strings -> int32 ids (first-seen order)."""
    table = {}
    outs = []
    for col in cols:
        ids = np.empty(len(col), np.int32)
        for i, v in enumerate(col):
            ids[i] = table.setdefault(v, len(table))
        outs.append(ids)
    return outs, len(table)
