"""Host-side passes of the front door, in the native library (include/ivjoin.h, "host-side helpers"; csrc/host_frontdoor.hip.h).

The reference runs these inside its Rust executor (DataFusion: dictionary handling of the join key, the column gathers of
the renaming SELECT, /root/reference/src/operation.rs:272-301; the int32 coordinate limit, docs/features/operations.md:36-37).
Each wrapper hands the buffers of numpy / Arrow arrays to ONE threaded pass; ctypes releases the interpreter lock for the call.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

from ._engine import _check, load_library

THREADS = max(1, min(int(os.environ.get("IVJ_HOST_THREADS", "32")), os.cpu_count() or 1))
MAX_DICT = 4096


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def narrow_i32(a: np.ndarray) -> Tuple[np.ndarray, int, int]:
    """Integer column -> (int32 copy, min, max) in one pass; the caller checks the range."""
    L = load_library()
    a = np.ascontiguousarray(a)
    out = np.empty(len(a), np.int32)
    mn, mx = C.c_int64(0), C.c_int64(0)
    _check(L, L.ivj_host_narrow_i32(_ptr(a), a.dtype.itemsize, 1 if a.dtype.kind == "u" else 0, len(a), _ptr(out), C.byref(mn), C.byref(mx), THREADS),
           "ivj_host_narrow_i32")
    return out, mn.value, mx.value


def encode_utf8(offsets: np.ndarray, data: Optional[np.ndarray], validity: Optional[np.ndarray], bit0: int, n: int, out: np.ndarray):
    """String column buffers -> ids written to ``out`` (int32, n), returns the rows that hold the dictionary values in
    first-occurrence order, or None when the column has more distinct values than the native encoder keeps."""
    L = load_library()
    rows = np.empty(MAX_DICT, np.int64)
    nv = C.c_int32(0)
    rc = L.ivj_host_encode_utf8(_ptr(offsets), offsets.dtype.itemsize, _ptr(data) if data is not None and len(data) else None,
                                _ptr(validity) if validity is not None else None, int(bit0), int(n), _ptr(out), _ptr(rows), MAX_DICT,
                                C.byref(nv), THREADS)
    if rc == -4:                                             # IVJ_ECAPACITY
        return None
    _check(L, rc, "ivj_host_encode_utf8")
    return rows[:nv.value].copy()


def encode_object_pointers(col: np.ndarray, out: np.ndarray):
    """Object-dtype numpy column -> ids (written to ``out``) by object IDENTITY; returns the rows holding the distinct objects in
    first-occurrence order, or None when there are more of them than the native encoder keeps."""
    L = load_library()
    assert col.dtype == object and col.flags.c_contiguous
    rows = np.empty(MAX_DICT, np.int64)
    nv = C.c_int32(0)
    rc = L.ivj_host_encode_keys64(C.c_void_p(col.ctypes.data), len(col), _ptr(out), _ptr(rows), MAX_DICT, C.byref(nv), THREADS)
    if rc == -4:
        return None
    _check(L, rc, "ivj_host_encode_keys64")
    return rows[:nv.value].copy()


def remap_i32(idx: np.ndarray, table: np.ndarray, out: np.ndarray, seen: Optional[np.ndarray] = None) -> None:
    """out[i] = table[idx[i]] (-1 for a negative index); seen[v] |= 1 for every table slot used."""
    L = load_library()
    idx = np.ascontiguousarray(idx)
    table = np.ascontiguousarray(table, np.int32)
    if seen is None:
        seen = np.zeros(max(len(table), 1), np.uint8)
    _check(L, L.ivj_host_remap_i32(_ptr(idx), idx.dtype.itemsize, len(idx), _ptr(table), len(table), _ptr(out), _ptr(seen), THREADS), "ivj_host_remap_i32")


def take(src: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """src[idx] for a contiguous 4- or 8-byte column (a negative index yields 0)."""
    L = load_library()
    out = np.empty(len(idx), src.dtype)
    _check(L, L.ivj_host_take(_ptr(src), src.dtype.itemsize, len(src), _ptr(idx), len(idx), _ptr(out), THREADS), "ivj_host_take")
    return out


def scatter(dst: np.ndarray, idx: np.ndarray, src: np.ndarray, remap: np.ndarray = None):
    """dst[idx] = src (rows of dst / src along axis 0; distinct indices), threaded and outside the GIL: the mirror of ``take``.
    remap (int32 values only): stores remap[v] for v >= 0 and -1 otherwise."""
    L = load_library()
    src = np.ascontiguousarray(src, dst.dtype)
    idx = np.ascontiguousarray(idx, np.int32)
    assert dst.flags.c_contiguous and len(src) == len(idx) and src.shape[1:] == dst.shape[1:]
    row_bytes = dst.dtype.itemsize * int(np.prod(dst.shape[1:], dtype=np.int64))
    rm = None if remap is None else np.ascontiguousarray(remap, np.int32)
    _check(L, L.ivj_host_scatter(_ptr(src), row_bytes, len(idx), _ptr(idx), len(dst), _ptr(dst), None if rm is None else _ptr(rm), 0 if rm is None else len(rm), THREADS),
           "ivj_host_scatter")


def widen_i64(src: np.ndarray) -> np.ndarray:
    L = load_library()
    out = np.empty(len(src), np.int64)
    _check(L, L.ivj_host_widen_i32(_ptr(src), len(src), _ptr(out), THREADS), "ivj_host_widen_i32")
    return out


def contig_hist(contig: np.ndarray, n_contigs: int) -> np.ndarray:
    """Rows per contig (ids outside [0, n_contigs) are not counted): one threaded pass."""
    L = load_library()
    contig = np.ascontiguousarray(contig, np.int32)
    out = np.zeros(max(int(n_contigs), 0), np.int64)
    _check(L, L.ivj_host_contig_hist(_ptr(contig), len(contig), int(n_contigs), _ptr(out), THREADS), "ivj_host_contig_hist")
    return out


def shard_by_owner(side, owner: np.ndarray, world: int):
    """(contig, start, end) int32 columns + owner[contig] -> per rank ((contig, start, end), global rows): one counting and one
    placing pass over the rows for ALL ranks (ivj_host_shard), input order kept inside a rank."""
    L = load_library()
    c, s, e = (np.ascontiguousarray(a, np.int32) for a in side)
    owner = np.ascontiguousarray(owner, np.int32)
    n, nc = len(c), len(owner)
    counts = np.zeros(world, np.int64)
    _check(L, L.ivj_host_shard(_ptr(c), _ptr(s), _ptr(e), n, _ptr(owner), nc, int(world), _ptr(counts), None, None, None, None, THREADS), "ivj_host_shard")
    outs = [[np.empty(int(counts[r]), np.int32) for r in range(world)] for _ in range(4)]
    arrs = [(C.c_void_p * world)(*[a.ctypes.data for a in col]) for col in outs]
    _check(L, L.ivj_host_shard(_ptr(c), _ptr(s), _ptr(e), n, _ptr(owner), nc, int(world), _ptr(counts), arrs[0], arrs[1], arrs[2], arrs[3], THREADS),
           "ivj_host_shard")
    return [((outs[0][r], outs[1][r], outs[2][r]), outs[3][r]) for r in range(world)]
