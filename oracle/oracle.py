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
        L.orc_overlap_tree.restype = C.c_int64
        L.orc_overlap_tree.argtypes = [C.c_void_p, P, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.orc_overlap_baseline.restype = C.c_int64
        L.orc_overlap_baseline.argtypes = [C.c_void_p, P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        L.orc_nearest_fast.restype = None
        L.orc_nearest_fast.argtypes = [C.c_void_p, P, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_set_threads.restype = None
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_place_i32.restype = None
        L.orc_place_i32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
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


def overlap_tree(ix: Index, probe: Side, strict: bool, threads: int = 0, count_only: bool = False):
    """Same pairs, same order as overlap_fast, through the implicit augmented interval tree (the stand-in for the
    reference's COITrees index)."""
    L = lib()
    n = L.orc_overlap_tree(ix.h, probe.ref(), int(strict), None, None, 0, threads)
    if count_only:
        return n
    p = np.empty(n, np.int32)
    b = np.empty(n, np.int32)
    L.orc_overlap_tree(ix.h, probe.ref(), int(strict), p.ctypes.data, b.ctypes.data, n, threads)
    return p, b


def overlap_baseline(ix: Index, probe: Side, strict: bool, threads: int, use_tree: bool = False, sort_chunks: bool = False):
    """Timed CPU baseline (bench.py): ONE call, one pass over the probe rows, pairs appended to recycled
    per-thread batches.  -> (number of pairs, sum of the emitted build rows)."""
    cs = C.c_int64(0)
    n = lib().orc_overlap_baseline(ix.h, probe.ref(), int(strict), int(threads), int(use_tree), int(sort_chunks), C.byref(cs))
    return int(n), int(cs.value)


def set_threads(n: int) -> None:
    """Threads of the index build's parallel passes (the other entry points take ``threads`` themselves)."""
    lib().orc_set_threads(int(n))


def placed(side: "Side", threads: int = 0) -> "Side":
    """Copy of a side whose pages are first touched by the threads that read them in overlap_baseline (orc_place_i32): NUMA
    placement for the all-core CPU baseline of bench.py; never inside a timed region."""
    cols = []
    for a in (side.contig, side.start, side.end):
        d = np.empty(len(a), np.int32)
        lib().orc_place_i32(a.ctypes.data, d.ctypes.data, len(a), int(threads))
        cols.append(d)
    return Side(*cols)


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


# ---------------------------------------------------------------------------------------------
# Sort-scan operations: merge / cluster / coverage / complement /
# subtract.  The arithmetic lives in the un-vendored crate behind MergeProvider / ClusterProvider /
# CountOverlapsProvider(coverage=true) / ComplementProvider / SubtractProvider
# (/root/reference/src/operation.rs:352-510, 306-350); what is restated here is the behaviour the
# reference's own tests pin:
#   merge ...... tests/_expected.py:174-181 (0-based: bookended intervals are NOT merged at
#                min_dist=0, i.e. pb.merge(min_dist=0) == bioframe.merge(min_dist=None),
#                tests/test_bioframe.py:120-124) and EXPECTED_MERGE of
#                tests/test_partitioned_range_operation_regressions.py:24-31
#   cluster .... EXPECTED_CLUSTER (:49-59); ids count clusters in (chrom, start) order
#                (tests/test_bioframe.py:398-419 compares the id column with bioframe's)
#   complement . EXPECTED_COMPLEMENT (:33-39), subtract: EXPECTED_SUBTRACT (:41-47)
#   coverage ... bases of every df1 interval covered by the union of df2 (tests/test_bioframe.py:302-340)
#   Weak vs Strict .. merge of adjacent intervals (2 rows 0-based, 1 row 1-based) and coverage of [100,200] by
#                [200,300] (0 / 1 positions): tests/test_coordinate_system_metadata.py:1032-1055, 1577-1623
# PARITY UNPINNED in the reference (no test fixes it; the choices below follow the Strict/Weak
# definition of range_op.py:75-84): min_dist > 0, cluster / complement / subtract under Weak
# (1-based closed) coordinates, rows with start > end, rows with a null chrom.
# Rule used throughout: a Weak (closed) interval [s, e] is the half-open interval [s, e + 1).

def _half_open_end(end: np.ndarray, strict: bool) -> np.ndarray:
    return end.astype(np.int64) + (0 if strict else 1)


def np_cluster(side: Side, strict: bool, min_dist: int = 0):
    """-> (cluster id per input row, cluster_start, cluster_end per input row, merged table
    (contig, start, end, n_intervals) in (contig id, start) order).  A row joins the running
    cluster iff start (<) running max end + min_dist, (<) being < for Strict and <= for Weak."""
    c, s, e = side.contig, side.start, side.end
    n = len(c)
    order = np.lexsort((np.arange(n), s, c))
    cid = np.empty(n, np.int64)
    cs = np.empty(n, np.int64)
    ce = np.empty(n, np.int64)
    m_c, m_s, m_e, m_n = [], [], [], []
    cur = -1
    cur_c = None
    cur_end = 0
    first = 0
    members = []
    for p in order:
        new = cur_c is None or c[p] != cur_c or not ((int(s[p]) < cur_end + min_dist) if strict else (int(s[p]) <= cur_end + min_dist))
        if new:
            if members:
                for r in members:
                    cid[r], cs[r], ce[r] = cur, first, cur_end
                m_c.append(cur_c); m_s.append(first); m_e.append(cur_end); m_n.append(len(members))
            cur += 1
            cur_c, first, cur_end, members = c[p], int(s[p]), int(e[p]), []
        cur_end = max(cur_end, int(e[p]))
        members.append(p)
    if members:
        for r in members:
            cid[r], cs[r], ce[r] = cur, first, cur_end
        m_c.append(cur_c); m_s.append(first); m_e.append(cur_end); m_n.append(len(members))
    merged = (np.array(m_c, np.int32), np.array(m_s, np.int64), np.array(m_e, np.int64), np.array(m_n, np.int64))
    return cid, cs, ce, merged


def np_coverage_brute(probe: Side, build: Side, strict: bool) -> np.ndarray:
    """Definition: for every probe row the number of integer positions of the probe interval that lie in
    at least one build interval of the same contig (positions of [s, e) for Strict, of [s, e] for Weak).
    O(Np * Nb) interval clipping + a union by sorting: small inputs only."""
    out = np.zeros(len(probe.contig), np.int64)
    be_all = _half_open_end(build.end, strict)
    for i in range(len(probe.contig)):
        qs, qe = int(probe.start[i]), int(probe.end[i]) + (0 if strict else 1)
        sel = build.contig == probe.contig[i]
        if probe.contig[i] < 0 or not sel.any() or qe <= qs:
            continue
        lo = np.maximum(build.start[sel].astype(np.int64), qs)
        hi = np.minimum(be_all[sel], qe)
        keep = hi > lo
        if not keep.any():
            continue
        lo, hi = lo[keep], hi[keep]
        o = np.argsort(lo, kind="stable")
        lo, hi = lo[o], hi[o]
        run = np.maximum.accumulate(hi)
        prev = np.concatenate([[lo[0]], run[:-1]])
        out[i] = int(np.maximum(hi - np.maximum(lo, prev), 0).sum())
    return out


def np_coverage_fast(probe: Side, build: Side, strict: bool) -> np.ndarray:
    """Same result through merged (disjoint) build intervals + prefix sums of their lengths."""
    mc, ms, me, _ = _merged_half_open(build, strict)
    out = np.zeros(len(probe.contig), np.int64)
    for c in np.unique(mc):
        sel = mc == c
        s_, e_ = ms[sel], me[sel]
        ln = np.maximum(e_ - s_, 0)
        pl = np.concatenate([[0], np.cumsum(ln)])
        rows = np.nonzero(probe.contig == c)[0]
        qs = probe.start[rows].astype(np.int64)
        qe = _half_open_end(probe.end[rows], strict)
        first = np.searchsorted(e_, qs, side="right")        # first merged interval with end > qs
        last = np.searchsorted(s_, qe, side="left")          # first merged interval with start >= qe
        ok = (last > first) & (qe > qs)
        f = np.minimum(first, len(s_) - 1)
        l = np.maximum(last - 1, 0)
        full = pl[np.maximum(last, first)] - pl[first]
        clip_l = np.maximum(qs - s_[f], 0)
        clip_r = np.maximum(e_[l] - qe, 0)
        # clipping never removes more than the end intervals hold
        cov = full - np.minimum(clip_l, ln[f]) - np.minimum(clip_r, ln[l])
        single = ok & (f == l)
        cov = np.where(single, np.maximum(np.minimum(e_[f], qe) - np.maximum(s_[f], qs), 0), cov)
        out[rows] = np.where(ok, np.maximum(cov, 0), 0)
    return out


def _merged_half_open(build: Side, strict: bool):
    """Disjoint union of the build intervals as half-open (contig, start, end) arrays, (contig, start) order."""
    c, s = build.contig, build.start.astype(np.int64)
    e = _half_open_end(build.end, strict)
    keep = (c >= 0) & (e > s)                             # rows that hold no position cover nothing
    c, s, e = c[keep], s[keep], e[keep]
    order = np.lexsort((s, c))
    c, s, e = c[order], s[order], e[order]
    mc, ms, me = [], [], []
    for i in range(len(c)):
        if mc and mc[-1] == c[i] and s[i] <= me[-1]:     # bookended intervals leave no gap between them
            me[-1] = max(me[-1], int(e[i]))
        else:
            mc.append(c[i]); ms.append(int(s[i])); me.append(int(e[i]))
    return np.array(mc, np.int32), np.array(ms, np.int64), np.array(me, np.int64), None


def _minus_union(c, s, e, mc, ms, me):
    """pieces of the half-open rows (c, s, e) outside the disjoint sorted union (mc, ms, me) -> [(row, start, end)]"""
    by_contig = {int(k): (ms[mc == k], me[mc == k]) for k in np.unique(mc)}
    out = []
    for i in range(len(c)):
        ls, le = int(s[i]), int(e[i])
        if le <= ls:
            continue
        us, ue = by_contig.get(int(c[i]), (None, None))
        cur = ls
        if us is not None:
            first = int(np.searchsorted(ue, ls, side="right"))     # first union interval with end > ls
            last = int(np.searchsorted(us, le, side="left"))       # first union interval with start >= le
            for j in range(first, last):
                if us[j] > cur:
                    out.append((i, cur, int(us[j])))
                cur = max(cur, int(ue[j]))
        if cur < le:
            out.append((i, cur, le))
    return np.array(out, np.int64).reshape(-1, 3)


def np_complement(side: Side, view: Side, strict: bool):
    """Gaps of the union of ``side`` inside every view interval -> (contig, start, end), in view order then by
    position.  Weak (closed) coordinates: the gap between [a, b] and [c, d] is [b + 1, c - 1]."""
    mc, ms, me, _ = _merged_half_open(side, strict)
    arr = _minus_union(view.contig, view.start.astype(np.int64), _half_open_end(view.end, strict), mc, ms, me)
    return view.contig[arr[:, 0]].astype(np.int32), arr[:, 1], arr[:, 2] - (0 if strict else 1)


def np_subtract(left: Side, right: Side, strict: bool):
    """Every left interval minus the union of the right intervals of its contig -> (left row, start, end) pieces,
    left-row order then by position (a fully covered left row yields nothing)."""
    mc, ms, me, _ = _merged_half_open(right, strict)
    arr = _minus_union(left.contig, left.start.astype(np.int64), _half_open_end(left.end, strict), mc, ms, me)
    return arr[:, 0].astype(np.int32), arr[:, 1], arr[:, 2] - (0 if strict else 1)
