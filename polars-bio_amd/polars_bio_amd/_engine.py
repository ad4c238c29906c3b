"""ctypes binding of libivjoin_hip.so (include/ivjoin.h).

This is the only place the package touches the native library.  There is no
CPU fallback: if the library is missing or no MI355X is usable the engine
raises -- it never routes through the oracle or numpy.

Replaces the reference's PyO3 entry point
``polars_bio.polars_bio.range_operation_frame`` (/root/reference/src/lib.rs:79-145).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libivjoin_hip.so")

FILTER_WEAK = 0    # FilterOp.Weak   (1-based closed)    src/option.rs:95-100
FILTER_STRICT = 1  # FilterOp.Strict (0-based half-open)

# every symbol include/ivjoin.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "ivj_last_error", "ivj_version", "ivj_abi_version", "ivj_device_count", "ivj_host_mem_available", "ivj_ctx_create", "ivj_ctx_destroy",
    "ivj_ctx_set_stream", "ivj_ctx_sync", "ivj_ctx_enable_timing", "ivj_ctx_get_timings", "ivj_ctx_profile_mark",
    "ivj_overlap", "ivj_pairs_free", "ivj_count_overlaps", "ivj_nearest",
    "ivj_index_build_dev", "ivj_index_free", "ivj_overlap_count_dev", "ivj_overlap_fill_dev", "ivj_overlap_fused_dev",
    "ivj_count_overlaps_dev", "ivj_nearest_dev",
    "ivj_side_from_arrow", "ivj_materialize_dev", "ivj_overlap_fused_rows_dev", "ivj_take_dev", "ivj_take", "ivj_overlap_rows", "ivj_rows_free", "ivj_rows_export_arrow",
    "ivj_subtract", "ivj_complement", "ivj_pieces_free", "ivj_subtract_dev",
    "ivj_merge", "ivj_merged_free", "ivj_cluster", "ivj_coverage", "ivj_cluster_dev", "ivj_merge_dev", "ivj_coverage_dev",
    "ivj_stream_open", "ivj_stream_submit", "ivj_stream_flush", "ivj_stream_close",
    "ivj_dev_alloc", "ivj_dev_free", "ivj_memcpy_h2d", "ivj_memcpy_d2h",
    "ivj_comm_unique_id", "ivj_comm_create", "ivj_comm_create_local", "ivj_comm_destroy", "ivj_comm_info",
    "ivj_allgather_counts", "ivj_allgatherv_dev", "ivj_overlap_allgather_dev",
    "ivj_count_overlaps_allgather_dev", "ivj_nearest_allgather_dev",
    "ivj_overlap_arrow_stream", "ivj_count_overlaps_arrow_stream", "ivj_nearest_arrow_stream", "ivj_arrow_encode_keys", "ivj_arrow_keys_free",
    "ivj_overlap_arrow_stream_lazy", "ivj_count_overlaps_arrow_stream_lazy", "ivj_nearest_arrow_stream_lazy",
    "ivj_arrow_take_stream",
    "ivj_host_shard", "ivj_host_contig_hist",
    "ivj_host_narrow_i32", "ivj_host_encode_utf8", "ivj_host_encode_keys64", "ivj_host_remap_i32", "ivj_host_take", "ivj_host_scatter", "ivj_host_widen_i32",
]

ABI_VERSION = 5            # include/ivjoin.h: IVJ_ABI_VERSION (struct layouts and signatures this binding assumes)
STREAM_OVERLAP, STREAM_COUNT, STREAM_NEAREST = 0, 1, 2

ROW_COLUMNS = ("probe_idx", "build_idx", "contig", "start_1", "end_1", "start_2", "end_2")


IVJ_ECAPACITY, IVJ_ESTATE, IVJ_EPEER = -4, -5, -6


class EngineError(RuntimeError):
    """HIP / engine failure (the reference surfaces these as PanicException).  ``code`` = the IVJ_E* status."""
    code = 0


class PeerError(EngineError):
    """A multi-rank call completed on this rank but ANOTHER rank failed (IVJ_EPEER): the gathered result is incomplete."""


class _Side(C.Structure):
    _fields_ = [("contig", C.c_void_p), ("start", C.c_void_p), ("end", C.c_void_p), ("n", C.c_int64),
                ("row_id", C.c_void_p)]


class _ArrowStream(C.Structure):
    """struct ArrowArrayStream (Arrow C stream interface): five pointers."""
    _fields_ = [("get_schema", C.c_void_p), ("get_next", C.c_void_p), ("get_last_error", C.c_void_p), ("release", C.c_void_p),
                ("private_data", C.c_void_p)]


class _ArrowKeys(C.Structure):
    _fields_ = [("n1", C.c_int64), ("n2", C.c_int64), ("contig1", C.c_void_p), ("start1", C.c_void_p), ("end1", C.c_void_p),
                ("contig2", C.c_void_p), ("start2", C.c_void_p), ("end2", C.c_void_p), ("n_contigs", C.c_int32),
                ("name_offsets", C.c_void_p), ("name_bytes", C.c_void_p)]


class _Opts(C.Structure):
    _fields_ = [("filter_op", C.c_int32), ("n_contigs", C.c_int32), ("nearest_k", C.c_int32),
                ("include_overlaps", C.c_int32), ("partition_mode", C.c_int32), ("table_mode", C.c_int32), ("slice_rows", C.c_int32), ("slice_chunk", C.c_int32),
                ("deterministic", C.c_int32)]


class _Pairs(C.Structure):
    _fields_ = [("n_pairs", C.c_int64), ("probe_idx", C.POINTER(C.c_int32)), ("build_idx", C.POINTER(C.c_int32))]


class _Rows(C.Structure):
    _fields_ = [("n_pairs", C.c_int64)] + [(name, C.POINTER(C.c_int32)) for name in
                                            ("probe_idx", "build_idx", "contig", "start_1", "end_1", "start_2", "end_2")]


class _Merged(C.Structure):
    _fields_ = [("n", C.c_int64), ("contig", C.POINTER(C.c_int32)), ("start", C.POINTER(C.c_int32)), ("end", C.POINTER(C.c_int32)),
                ("n_intervals", C.POINTER(C.c_int64))]


class _Pieces(C.Structure):
    _fields_ = [("n", C.c_int64), ("row", C.POINTER(C.c_int32)), ("start", C.POINTER(C.c_int32)), ("end", C.POINTER(C.c_int32))]


class _ArrowSchema(C.Structure):     # Arrow C Data Interface, opaque to Python: only its address is handed on
    _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
                ("n_children", C.c_int64), ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p),
                ("private_data", C.c_void_p)]


class _ArrowArray(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
                ("n_children", C.c_int64), ("buffers", C.c_void_p), ("children", C.c_void_p), ("dictionary", C.c_void_p),
                ("release", C.c_void_p), ("private_data", C.c_void_p)]


class _StreamResult(C.Structure):
    _fields_ = [("batch", C.c_int64), ("n_probe", C.c_int64), ("n", C.c_int64), ("probe_idx", C.POINTER(C.c_int32)),
                ("build_idx", C.POINTER(C.c_int32)), ("counts", C.POINTER(C.c_int64)), ("dist", C.POINTER(C.c_int64)),
                ("n_found", C.POINTER(C.c_int32))]


class _Timing(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int32), ("ms", C.c_float)]


_lib = None
_lib_lock = threading.Lock()


def load_library() -> C.CDLL:
    """dlopen the in-tree HIP library; raise loudly when it is absent."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise EngineError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        P, O = C.POINTER(_Side), C.POINTER(_Opts)
        vp = C.c_void_p
        L.ivj_last_error.restype = C.c_char_p
        if not hasattr(L, "ivj_abi_version") or L.ivj_abi_version() != ABI_VERSION:
            raise EngineError(f"{LIB_PATH} speaks ABI version {L.ivj_abi_version() if hasattr(L, 'ivj_abi_version') else '?'}, this binding was written "
                              f"against {ABI_VERSION} (include/ivjoin.h: IVJ_ABI_VERSION): rebuild the library")
        L.ivj_version.restype = C.c_char_p
        L.ivj_host_mem_available.restype = C.c_int64
        L.ivj_device_count.argtypes = [C.POINTER(C.c_int)]
        L.ivj_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.ivj_ctx_destroy.argtypes = [vp]
        L.ivj_ctx_destroy.restype = None
        L.ivj_ctx_set_stream.argtypes = [vp, vp]
        L.ivj_ctx_sync.argtypes = [vp]
        L.ivj_ctx_enable_timing.argtypes = [vp, C.c_int]
        L.ivj_ctx_profile_mark.argtypes = [vp]
        L.ivj_ctx_get_timings.argtypes = [vp, C.POINTER(_Timing), C.c_int, C.POINTER(C.c_int)]
        L.ivj_overlap.argtypes = [vp, P, P, O, C.POINTER(_Pairs)]
        L.ivj_pairs_free.argtypes = [C.POINTER(_Pairs)]
        L.ivj_pairs_free.restype = None
        L.ivj_count_overlaps.argtypes = [vp, P, P, O, vp]
        L.ivj_nearest.argtypes = [vp, P, P, O, vp, vp, vp]
        L.ivj_index_build_dev.argtypes = [vp, P, O, C.c_int, C.POINTER(vp)]
        L.ivj_index_free.argtypes = [vp]
        L.ivj_index_free.restype = None
        L.ivj_overlap_count_dev.argtypes = [vp, vp, P, O, C.POINTER(C.c_int64)]
        L.ivj_overlap_fill_dev.argtypes = [vp, vp, P, O, vp, vp, C.c_int64]
        L.ivj_overlap_fused_dev.argtypes = [vp, vp, P, O, vp, vp, C.c_int64, C.POINTER(C.c_int64)]
        L.ivj_count_overlaps_dev.argtypes = [vp, vp, P, O, vp]
        L.ivj_nearest_dev.argtypes = [vp, vp, P, O, vp, vp, vp]
        L.ivj_materialize_dev.argtypes = [vp, P, P, C.POINTER(_Rows)]
        L.ivj_overlap_fused_rows_dev.argtypes = [vp, vp, P, O, C.POINTER(_Rows), C.POINTER(C.c_int64)]
        L.ivj_take_dev.argtypes = [vp, vp, C.c_int32, vp, C.c_int64, vp, vp]
        L.ivj_take.argtypes = [vp, vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp]
        L.ivj_overlap_rows.argtypes = [vp, P, P, O, C.POINTER(_Rows)]
        L.ivj_rows_free.argtypes = [C.POINTER(_Rows)]
        L.ivj_rows_free.restype = None
        L.ivj_rows_export_arrow.argtypes = [C.POINTER(_Rows), vp, vp]
        L.ivj_side_from_arrow.argtypes = [vp, vp, P]
        L.ivj_subtract.argtypes = [vp, P, P, O, C.POINTER(_Pieces)]
        L.ivj_complement.argtypes = [vp, P, P, O, C.POINTER(_Pieces)]
        L.ivj_pieces_free.argtypes = [C.POINTER(_Pieces)]
        L.ivj_pieces_free.restype = None
        L.ivj_subtract_dev.argtypes = [vp, vp, P, O, C.c_int64, vp, vp, vp, C.POINTER(C.c_int64)]
        L.ivj_merge.argtypes = [vp, P, O, C.c_int64, C.POINTER(_Merged)]
        L.ivj_merged_free.argtypes = [C.POINTER(_Merged)]
        L.ivj_merged_free.restype = None
        L.ivj_cluster.argtypes = [vp, P, O, C.c_int64, vp, vp, vp, C.POINTER(C.c_int64)]
        L.ivj_coverage.argtypes = [vp, P, P, O, vp]
        L.ivj_cluster_dev.argtypes = [vp, vp, O, C.c_int64, vp, vp, vp, C.POINTER(C.c_int64)]
        L.ivj_merge_dev.argtypes = [vp, vp, O, C.c_int64, C.c_int64, vp, vp, vp, vp, C.POINTER(C.c_int64)]
        L.ivj_coverage_dev.argtypes = [vp, vp, P, O, vp]
        L.ivj_stream_open.argtypes = [vp, P, O, C.c_int, C.c_int64, C.POINTER(vp)]
        L.ivj_stream_submit.argtypes = [vp, P, C.POINTER(_StreamResult)]
        L.ivj_stream_flush.argtypes = [vp, C.POINTER(_StreamResult)]
        L.ivj_stream_close.argtypes = [vp]
        L.ivj_stream_close.restype = None
        L.ivj_dev_alloc.argtypes = [vp, C.c_int64, C.POINTER(vp)]
        L.ivj_dev_free.argtypes = [vp, vp]
        L.ivj_memcpy_h2d.argtypes = [vp, vp, vp, C.c_int64]
        L.ivj_memcpy_d2h.argtypes = [vp, vp, vp, C.c_int64]
        L.ivj_comm_unique_id.argtypes = [vp]
        L.ivj_comm_create.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
        L.ivj_comm_create_local.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp)]
        L.ivj_comm_destroy.argtypes = [vp]
        L.ivj_comm_destroy.restype = None
        L.ivj_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ivj_allgather_counts.argtypes = [vp, C.c_int64, C.POINTER(C.c_int64)]
        L.ivj_allgatherv_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.POINTER(C.c_int64)]
        L.ivj_overlap_allgather_dev.argtypes = [vp, vp, P, O, C.c_int, vp, vp, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.ivj_count_overlaps_allgather_dev.argtypes = [vp, vp, P, O, C.c_int64, vp]
        L.ivj_nearest_allgather_dev.argtypes = [vp, vp, P, O, C.c_int64, vp, vp, vp]
        L.ivj_host_narrow_i32.argtypes = [vp, C.c_int32, C.c_int32, C.c_int64, vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32]
        L.ivj_host_encode_utf8.argtypes = [vp, C.c_int32, vp, vp, C.c_int64, C.c_int64, vp, vp, C.c_int32, C.POINTER(C.c_int32), C.c_int32]
        L.ivj_host_encode_keys64.argtypes = [vp, C.c_int64, vp, vp, C.c_int32, C.POINTER(C.c_int32), C.c_int32]
        L.ivj_host_remap_i32.argtypes = [vp, C.c_int32, C.c_int64, vp, C.c_int64, vp, vp, C.c_int32]
        L.ivj_host_take.argtypes = [vp, C.c_int32, C.c_int64, vp, C.c_int64, vp, C.c_int32]
        L.ivj_host_scatter.argtypes = [vp, C.c_int32, C.c_int64, vp, C.c_int64, vp, vp, C.c_int64, C.c_int32]
        L.ivj_host_widen_i32.argtypes = [vp, C.c_int64, vp, C.c_int32]
        L.ivj_host_shard.argtypes = [vp, vp, vp, C.c_int64, vp, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, C.c_int32]
        L.ivj_host_contig_hist.argtypes = [vp, C.c_int64, C.c_int32, vp, C.c_int32]
        names3 = C.POINTER(C.c_char_p)
        L.ivj_overlap_arrow_stream.argtypes = [vp, vp, vp, names3, names3, O, C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, vp]
        L.ivj_count_overlaps_arrow_stream.argtypes = [vp, vp, vp, names3, names3, O, C.c_char_p, C.c_int64, C.c_int64, vp]
        L.ivj_nearest_arrow_stream.argtypes = [vp, vp, vp, names3, names3, O, C.c_char_p, C.c_char_p, C.c_int32, C.c_int64, C.c_int64, vp]
        L.ivj_overlap_arrow_stream_lazy.argtypes = [vp, vp, vp, names3, names3, O, C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int64, vp]
        L.ivj_count_overlaps_arrow_stream_lazy.argtypes = [vp, vp, vp, names3, names3, O, C.c_char_p, C.c_int64, C.c_int64, C.c_int64, vp]
        L.ivj_nearest_arrow_stream_lazy.argtypes = [vp, vp, vp, names3, names3, O, C.c_char_p, C.c_char_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64, vp]
        L.ivj_arrow_encode_keys.argtypes = [vp, vp, names3, names3, C.POINTER(_ArrowKeys)]
        L.ivj_arrow_keys_free.argtypes = [C.POINTER(_ArrowKeys)]
        L.ivj_arrow_keys_free.restype = None
        L.ivj_arrow_take_stream.argtypes = [vp, vp, C.c_int64, C.c_int64, vp]
        _lib = L
        return L


def _check(L, rc: int, what: str):
    if rc != 0:
        msg = L.ivj_last_error()
        err = (PeerError if rc == IVJ_EPEER else EngineError)(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
        err.code = rc
        raise err


def device_count() -> int:
    L = load_library()
    n = C.c_int(0)
    rc = L.ivj_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def _i32(a) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype != np.int32 or not a.flags.c_contiguous:
        a = np.ascontiguousarray(a, dtype=np.int32)
    return a


def _host_side(contig, start, end) -> Tuple[_Side, tuple]:
    c, s, e = _i32(contig), _i32(start), _i32(end)
    if not (c.shape == s.shape == e.shape and c.ndim == 1):
        raise ValueError("contig/start/end must be 1-D arrays of equal length")
    return _Side(c.ctypes.data, s.ctypes.data, e.ctypes.data, c.shape[0], None), (c, s, e)


def side_from_arrow(batch) -> Tuple[_Side, tuple]:
    """pyarrow.RecordBatch / StructArray with int32 children contig / start / end -> ivj_side viewing the Arrow
    buffers (ivj_side_from_arrow, zero copy).  Returns (side, keep-alive); needs no device."""
    L = load_library()
    arr, sch = _ArrowArray(), _ArrowSchema()
    batch._export_to_c(C.addressof(arr), C.addressof(sch))
    side = _Side()
    _check(L, L.ivj_side_from_arrow(C.addressof(arr), C.addressof(sch), C.byref(side)), "ivj_side_from_arrow")
    return side, (batch, arr, sch)


def make_opts(strict: bool, n_contigs: int, k: int = 1, include_overlaps: bool = True, partition_mode: int = 0,
              table_mode: int = 0, slice_rows: int = 0, slice_chunk: int = 0, deterministic: bool = False) -> _Opts:
    o = _Opts()
    o.deterministic = 1 if deterministic else 0
    o.slice_rows = int(slice_rows)
    o.slice_chunk = int(slice_chunk)
    o.table_mode = int(table_mode)
    o.partition_mode = int(partition_mode)
    o.filter_op = FILTER_STRICT if strict else FILTER_WEAK
    o.n_contigs = int(n_contigs)
    o.nearest_k = int(k)
    o.include_overlaps = 1 if include_overlaps else 0
    return o


class _LockedLib:
    """The native library as one Engine sees it: every call is made under that engine's lock.

    An ivj_ctx is not thread-safe (include/ivjoin.h: arena, count -> fill state, pinned totals and the index
    cache are per context) and ctypes releases the GIL during a call, so two Python threads sharing an Engine
    would otherwise race inside the library.  Multi-call sequences (count -> fill, streaming tiles) take
    ``Engine.lock`` around the whole sequence as well (it is re-entrant)."""

    def __init__(self, lib: C.CDLL, lock):
        self._lib, self._lock = lib, lock

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        lock = self._lock

        def call(*args):
            with lock:
                return fn(*args)
        call.__name__ = name
        setattr(self, name, call)
        return call


class _PairsOwner:
    """Keeps an ivj_pairs result alive for the numpy arrays that view its buffers."""

    def __init__(self, lib, pairs):
        self.lib, self.pairs = lib, pairs

    def __del__(self):
        try:
            self.lib.ivj_pairs_free(C.byref(self.pairs))
        except Exception:
            pass


def _owned_view(ptr, n, owner) -> np.ndarray:
    """numpy view of `n` int32 at `ptr` whose base object keeps `owner` alive (views of the view chain back to it)."""
    buf = (C.c_int32 * n).from_address(C.addressof(ptr.contents))
    buf._owner = owner                       # the ctypes array is the numpy array's base: its attribute pins the owner
    return np.frombuffer(buf, dtype=np.int32)


class ProbeStream:
    """Streaming probe session (ivj_stream_*): the build side is indexed once, probe batches are submitted one at a time;
    every submit overlaps the H2D copy of its batch, the join of the previous batch and the D2H copy of the one before.

    submit() / flush() return None or a dict: ``batch`` (index of the batch the results belong to), ``n_probe`` and
        overlap         ``probe_idx`` (rows INSIDE that batch), ``build_idx``
        count_overlaps  ``counts``
        nearest         ``build_idx`` (n_probe x k, -1 = none), ``dist`` (n_probe x k), ``n_found``
    The arrays are copies unless ``copy=False`` (then they view the stream's pinned result slot and are valid until the
    next call on the stream)."""

    def __init__(self, engine: "Engine", build, strict: bool, n_contigs: int, op: int, max_batch_rows: int, k: int = 1,
                 include_overlaps: bool = True, partition_mode: int = 0, copy: bool = True):
        self.engine, self.op, self.k, self.copy = engine, int(op), int(k), copy
        self.opts = make_opts(strict, n_contigs, k, include_overlaps, partition_mode=partition_mode)
        bs, keep = _host_side(*build)
        h = C.c_void_p()
        _check(engine.L, engine.L.ivj_stream_open(engine.h, C.byref(bs), C.byref(self.opts), self.op, int(max_batch_rows), C.byref(h)),
               "ivj_stream_open")
        del keep
        self.h = h
        self.max_batch_rows = int(max_batch_rows)

    def _result(self, r: _StreamResult):
        if r.batch < 0:
            return None
        def arr(ptr, n, dtype, shape=None):
            if n == 0:
                a = np.empty(0, dtype)
            else:
                a = np.ctypeslib.as_array(ptr, shape=(n,))
                a = a.copy() if self.copy else a
            return a.reshape(shape) if shape else a
        out = {"batch": int(r.batch), "n_probe": int(r.n_probe)}
        n, k = int(r.n_probe), self.k
        if self.op == STREAM_OVERLAP:
            out["probe_idx"] = arr(r.probe_idx, int(r.n), np.int32)
            out["build_idx"] = arr(r.build_idx, int(r.n), np.int32)
        elif self.op == STREAM_COUNT:
            out["counts"] = arr(r.counts, n, np.int64)
        else:
            out["build_idx"] = arr(r.build_idx, n * k, np.int32, (n, k))
            out["dist"] = arr(r.dist, n * k, np.int64, (n, k))
            out["n_found"] = arr(r.n_found, n, np.int32)
        return out

    def submit(self, batch):
        """batch: (contig, start, end) int32 arrays, or an ivj_side made by side_from_arrow (zero copy from Arrow)."""
        keep = None
        if isinstance(batch, _Side):
            side = batch
        else:
            side, keep = _host_side(*batch)
        r = _StreamResult()
        _check(self.engine.L, self.engine.L.ivj_stream_submit(self.h, C.byref(side), C.byref(r)), "ivj_stream_submit")
        del keep
        return self._result(r)

    def flush(self):
        r = _StreamResult()
        _check(self.engine.L, self.engine.L.ivj_stream_flush(self.h, C.byref(r)), "ivj_stream_flush")
        return self._result(r)

    def close(self):
        if getattr(self, "h", None):
            self.engine.L.ivj_stream_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceIndex:
    """Sorted build side resident in HBM (ivj_index)."""

    def __init__(self, engine: "Engine", handle: int, n: int):
        self.engine, self.handle, self.n = engine, handle, n

    def close(self):
        if self.handle:
            self.engine.L.ivj_index_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One HIP device + stream + scratch arena (ivj_ctx)."""

    def __init__(self, device: int = 0):
        self.lock = threading.RLock()
        self.L = _LockedLib(load_library(), self.lock)
        h = C.c_void_p()
        _check(self.L, self.L.ivj_ctx_create(int(device), C.byref(h)), "ivj_ctx_create")
        self.h = h
        self.device = device

    def close(self):
        """Destroys the context.  Indexes built on it stay valid handles (the library detaches them) and are
        released by their own close()."""
        with self.lock:
            if getattr(self, "h", None):
                self.L.ivj_ctx_destroy(self.h)
                self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-buffer entry points (numpy in, numpy out) --------------------
    def overlap(self, probe, build, strict: bool, n_contigs: int, partition_mode: int = 0, table_mode: int = 0, slice_rows: int = 0,
                slice_chunk: int = 0, deterministic: bool = False):
        """probe/build: (contig_id, start, end) int32 arrays -> (probe_idx, build_idx).
        partition_mode: 0 auto, 1 256 buckets + window scan, 2 never (pairs then come in probe-row order), 6 index slices in LDS.
        deterministic: the slice path's output is identical from run to run (stable partition; ivj_opts.deterministic)."""
        ps, keep_p = _host_side(*probe)
        bs, keep_b = _host_side(*build)
        o = make_opts(strict, n_contigs, partition_mode=partition_mode, table_mode=table_mode, slice_rows=slice_rows, slice_chunk=slice_chunk,
                      deterministic=deterministic)
        out = _Pairs()
        _check(self.L, self.L.ivj_overlap(self.h, C.byref(ps), C.byref(bs), C.byref(o), C.byref(out)), "ivj_overlap")
        del keep_p, keep_b
        n = out.n_pairs
        if n == 0:
            self.L.ivj_pairs_free(C.byref(out))
            return np.empty(0, np.int32), np.empty(0, np.int32)
        # no copy: the arrays view the library's (pre-faulted, huge-page) result buffers, which are freed when the last
        # view of either array is gone
        owner = _PairsOwner(load_library(), out)
        return _owned_view(out.probe_idx, n, owner), _owned_view(out.build_idx, n, owner)

    def take_columns(self, idx: np.ndarray, columns, nullable: bool = False):
        """Arrow ``take`` of fixed-width host columns through HBM (ivj_take): ``columns`` = numpy arrays of 4- or 8-byte
        items sharing the int32 index column ``idx``.  -> list of (values, validity) with validity = None, or (nullable) a
        uint64 bitmap (Arrow layout) whose cleared bits mark the negative-index slots."""
        idx = np.ascontiguousarray(idx, np.int32)
        n, k = int(idx.shape[0]), len(columns)
        cols = [np.ascontiguousarray(c) for c in columns]
        for c in cols:
            if c.dtype.itemsize not in (4, 8) or c.ndim != 1:
                raise ValueError("take_columns: 1-d columns of 4- or 8-byte items only")
        outs = [np.empty(n, c.dtype) for c in cols]
        vals = [np.empty((n + 63) // 64, np.uint64) if nullable else None for _ in cols]
        if n == 0 or k == 0:
            return list(zip(outs, vals))
        VP = C.c_void_p * k
        src = VP(*[c.ctypes.data for c in cols])
        dst = VP(*[o.ctypes.data for o in outs])
        val = VP(*[(v.ctypes.data if v is not None else None) for v in vals])
        rows = (C.c_int64 * k)(*[int(c.shape[0]) for c in cols])
        eb = (C.c_int32 * k)(*[int(c.dtype.itemsize) for c in cols])
        _check(self.L, self.L.ivj_take(self.h, C.c_void_p(idx.ctypes.data), n, k, src, rows, eb, dst, val if nullable else None), "ivj_take")
        return list(zip(outs, vals))

    def overlap_rows(self, probe, build, strict: bool, n_contigs: int, partition_mode: int = 0, as_arrow: bool = False):
        """overlap + row materialisation on the device (ivj_overlap_rows): the pair indices AND the key
        columns of both sides for every pair.  -> dict of int32 numpy arrays keyed by ROW_COLUMNS, or
        (as_arrow=True) a pyarrow.RecordBatch that owns the library's host buffers through the Arrow C
        Data interface (ivj_rows_export_arrow; no copy)."""
        ps, keep_p = _host_side(*probe)
        bs, keep_b = _host_side(*build)
        o = make_opts(strict, n_contigs, partition_mode=partition_mode)
        out = _Rows()
        _check(self.L, self.L.ivj_overlap_rows(self.h, C.byref(ps), C.byref(bs), C.byref(o), C.byref(out)), "ivj_overlap_rows")
        del keep_p, keep_b
        try:
            if as_arrow:
                import pyarrow as pa
                arr, sch = _ArrowArray(), _ArrowSchema()
                _check(self.L, self.L.ivj_rows_export_arrow(C.byref(out), C.addressof(arr), C.addressof(sch)),
                       "ivj_rows_export_arrow")
                return pa.RecordBatch._import_from_c(C.addressof(arr), C.addressof(sch))
            n = out.n_pairs
            return {name: (np.ctypeslib.as_array(getattr(out, name), shape=(n,)).copy() if n else np.empty(0, np.int32))
                    for name in ROW_COLUMNS}
        finally:
            self.L.ivj_rows_free(C.byref(out))

    # ---- sort-scan family ---------------------------------------------
    def merge(self, frame, strict: bool, n_contigs: int, min_dist: int = 0):
        """pb.merge: -> (contig id, start, end, n_intervals) of the merged intervals, (contig id, start) order."""
        fs, keep = _host_side(*frame)
        o = make_opts(strict, n_contigs)
        out = _Merged()
        _check(self.L, self.L.ivj_merge(self.h, C.byref(fs), C.byref(o), int(min_dist), C.byref(out)), "ivj_merge")
        del keep
        try:
            n = out.n
            if n == 0:
                return np.empty(0, np.int32), np.empty(0, np.int32), np.empty(0, np.int32), np.empty(0, np.int64)
            return (np.ctypeslib.as_array(out.contig, shape=(n,)).copy(), np.ctypeslib.as_array(out.start, shape=(n,)).copy(),
                    np.ctypeslib.as_array(out.end, shape=(n,)).copy(), np.ctypeslib.as_array(out.n_intervals, shape=(n,)).copy())
        finally:
            self.L.ivj_merged_free(C.byref(out))

    def cluster(self, frame, strict: bool, n_contigs: int, min_dist: int = 0):
        """pb.cluster: -> (cluster id int64, cluster_start, cluster_end) per input row + number of clusters."""
        fs, keep = _host_side(*frame)
        o = make_opts(strict, n_contigs)
        cid = np.empty(fs.n, np.int64)
        cs = np.empty(fs.n, np.int32)
        ce = np.empty(fs.n, np.int32)
        ncl = C.c_int64(0)
        _check(self.L, self.L.ivj_cluster(self.h, C.byref(fs), C.byref(o), int(min_dist), cid.ctypes.data, cs.ctypes.data,
                                           ce.ctypes.data, C.byref(ncl)), "ivj_cluster")
        del keep
        return cid, cs, ce, ncl.value

    def coverage(self, probe, build, strict: bool, n_contigs: int, partition_mode: int = 0) -> np.ndarray:
        """pb.coverage: covered positions of every probe row (int64, probe order)."""
        ps, keep_p = _host_side(*probe)
        bs, keep_b = _host_side(*build)
        o = make_opts(strict, n_contigs, partition_mode=partition_mode)
        cov = np.empty(ps.n, np.int64)
        _check(self.L, self.L.ivj_coverage(self.h, C.byref(ps), C.byref(bs), C.byref(o), cov.ctypes.data), "ivj_coverage")
        del keep_p, keep_b
        return cov

    def _pieces(self, fn, name, a, b, strict, n_contigs, partition_mode=0):
        sa, keep_a = _host_side(*a)
        sb, keep_b = _host_side(*b)
        o = make_opts(strict, n_contigs, partition_mode=partition_mode)
        out = _Pieces()
        _check(self.L, fn(self.h, C.byref(sa), C.byref(sb), C.byref(o), C.byref(out)), name)
        del keep_a, keep_b
        try:
            n = out.n
            if n == 0:
                return np.empty(0, np.int32), np.empty(0, np.int32), np.empty(0, np.int32)
            return tuple(np.ctypeslib.as_array(getattr(out, k), shape=(n,)).copy() for k in ("row", "start", "end"))
        finally:
            self.L.ivj_pieces_free(C.byref(out))

    def subtract(self, left, right, strict: bool, n_contigs: int, partition_mode: int = 0):
        """pb.subtract: every left interval minus the union of the right ones -> (left row, start, end) pieces.
        partition_mode 0 auto / 1 bucket the left rows first / 2 never: same result, same order."""
        return self._pieces(self.L.ivj_subtract, "ivj_subtract", left, right, strict, n_contigs, partition_mode)

    def complement(self, frame, view, strict: bool, n_contigs: int, partition_mode: int = 0):
        """pb.complement: the gaps of ``frame`` inside every view interval -> (view row, start, end)."""
        return self._pieces(self.L.ivj_complement, "ivj_complement", frame, view, strict, n_contigs, partition_mode)

    def count_overlaps(self, probe, build, strict: bool, n_contigs: int, table_mode: int = 0, partition_mode: int = 0) -> np.ndarray:
        ps, keep_p = _host_side(*probe)
        bs, keep_b = _host_side(*build)
        o = make_opts(strict, n_contigs, table_mode=table_mode, partition_mode=partition_mode)
        counts = np.empty(ps.n, np.int64)
        _check(self.L, self.L.ivj_count_overlaps(self.h, C.byref(ps), C.byref(bs), C.byref(o), counts.ctypes.data),
               "ivj_count_overlaps")
        del keep_p, keep_b
        return counts

    def nearest(self, probe, build, strict: bool, n_contigs: int, k: int = 1, include_overlaps: bool = True,
                table_mode: int = 0, partition_mode: int = 0):
        """partition_mode 0 auto / 1 bucket the probe side first / 2 never: same result, probe order kept."""
        ps, keep_p = _host_side(*probe)
        bs, keep_b = _host_side(*build)
        if k < 1:
            raise ValueError("k must be >= 1")
        o = make_opts(strict, n_contigs, k, include_overlaps, table_mode=table_mode, partition_mode=partition_mode)
        idx = np.empty((ps.n, k), np.int32)
        dist = np.empty((ps.n, k), np.int64)
        nf = np.empty(ps.n, np.int32)
        _check(self.L, self.L.ivj_nearest(self.h, C.byref(ps), C.byref(bs), C.byref(o), idx.ctypes.data,
                                           dist.ctypes.data, nf.ctypes.data), "ivj_nearest")
        del keep_p, keep_b
        return idx, dist, nf

    # ---- device-resident entry points (raw device pointers) -----------------
    def set_stream(self, hip_stream: Optional[int]):
        """hipStream_t handle as an int (0 = the legacy default stream torch uses); None restores
        the engine's own stream."""
        h = C.c_void_p(-1) if hip_stream is None else C.c_void_p(hip_stream)
        _check(self.L, self.L.ivj_ctx_set_stream(self.h, h), "ivj_ctx_set_stream")

    def sync(self):
        _check(self.L, self.L.ivj_ctx_sync(self.h), "ivj_ctx_sync")

    def enable_timing(self, level: int = 2):
        """0 off, 1 probe kernels only, 2 every kernel (HIP events on the launch stream)."""
        _check(self.L, self.L.ivj_ctx_enable_timing(self.h, int(level)), "ivj_ctx_enable_timing")

    def profile_mark(self):
        """An empty marker kernel on the context's stream (step boundary for a profiler)."""
        _check(self.L, self.L.ivj_ctx_profile_mark(self.h), "ivj_ctx_profile_mark")

    def timings(self) -> dict:
        arr = (_Timing * 64)()
        n = C.c_int(0)
        _check(self.L, self.L.ivj_ctx_get_timings(self.h, arr, 64, C.byref(n)), "ivj_ctx_get_timings")
        return {arr[i].name.decode(): {"launches": arr[i].launches, "ms": arr[i].ms} for i in range(min(n.value, 64))}

    @staticmethod
    def dev_side(contig_ptr: int, start_ptr: int, end_ptr: int, n: int, row_id_ptr: int = 0) -> _Side:
        return _Side(contig_ptr or None, start_ptr or None, end_ptr or None, n, row_id_ptr or None)

    def index_build_dev(self, build: _Side, opts: _Opts, with_end_order: bool = False, sweep_only: bool = False) -> DeviceIndex:
        """sweep_only: index for merge_dev / cluster_dev only (no lookup tables)."""
        h = C.c_void_p()
        with_end_order = int(bool(with_end_order)) | (2 if sweep_only else 0)
        _check(self.L, self.L.ivj_index_build_dev(self.h, C.byref(build), C.byref(opts), int(with_end_order), C.byref(h)),
               "ivj_index_build_dev")
        return DeviceIndex(self, h, build.n)

    def overlap_count_dev(self, ix: DeviceIndex, probe: _Side, opts: _Opts) -> int:
        n = C.c_int64(0)
        _check(self.L, self.L.ivj_overlap_count_dev(self.h, ix.handle, C.byref(probe), C.byref(opts), C.byref(n)),
               "ivj_overlap_count_dev")
        return n.value

    def overlap_fill_dev(self, ix: DeviceIndex, probe: _Side, opts: _Opts, probe_idx_ptr: int, build_idx_ptr: int,
                         capacity: int):
        _check(self.L, self.L.ivj_overlap_fill_dev(self.h, ix.handle, C.byref(probe), C.byref(opts),
                                                    C.c_void_p(probe_idx_ptr), C.c_void_p(build_idx_ptr), capacity),
               "ivj_overlap_fill_dev")

    def overlap_fused_dev(self, ix: DeviceIndex, probe: _Side, opts: _Opts, probe_idx_ptr: int, build_idx_ptr: int,
                          capacity: int):
        """-> (n_pairs, fits).  fits=False: nothing usable was written, grow the buffers to n_pairs."""
        n = C.c_int64(0)
        rc = self.L.ivj_overlap_fused_dev(self.h, ix.handle, C.byref(probe), C.byref(opts), C.c_void_p(probe_idx_ptr),
                                          C.c_void_p(build_idx_ptr), capacity, C.byref(n))
        if rc == -4:      # IVJ_ECAPACITY
            return n.value, False
        _check(self.L, rc, "ivj_overlap_fused_dev")
        return n.value, True

    def materialize_dev(self, probe: _Side, build: _Side, n_pairs: int, probe_idx_ptr: int, build_idx_ptr: int,
                        contig_ptr: int = 0, start_1_ptr: int = 0, end_1_ptr: int = 0, start_2_ptr: int = 0, end_2_ptr: int = 0):
        """ivj_materialize_dev: gather the key columns of both sides for every pair (device pointers;
        0 skips a column)."""
        r = _Rows()
        r.n_pairs = int(n_pairs)
        for name, ptr in zip(ROW_COLUMNS, (probe_idx_ptr, build_idx_ptr, contig_ptr, start_1_ptr, end_1_ptr, start_2_ptr, end_2_ptr)):
            setattr(r, name, C.cast(C.c_void_p(ptr or None), C.POINTER(C.c_int32)))
        _check(self.L, self.L.ivj_materialize_dev(self.h, C.byref(probe), C.byref(build), C.byref(r)), "ivj_materialize_dev")

    def overlap_fused_rows_dev(self, ix: DeviceIndex, probe: _Side, opts: _Opts, capacity: int, probe_idx_ptr: int = 0,
                               build_idx_ptr: int = 0, contig_ptr: int = 0, start_1_ptr: int = 0, end_1_ptr: int = 0,
                               start_2_ptr: int = 0, end_2_ptr: int = 0):
        """ivj_overlap_fused_rows_dev: join + key-column materialisation in one pass.  -> (n_rows, fits)."""
        r = _Rows()
        r.n_pairs = int(capacity)
        for name, ptr in zip(ROW_COLUMNS, (probe_idx_ptr, build_idx_ptr, contig_ptr, start_1_ptr, end_1_ptr, start_2_ptr, end_2_ptr)):
            setattr(r, name, C.cast(C.c_void_p(ptr or None), C.POINTER(C.c_int32)))
        n = C.c_int64(0)
        rc = self.L.ivj_overlap_fused_rows_dev(self.h, ix.handle, C.byref(probe), C.byref(opts), C.byref(r), C.byref(n))
        if rc == -4:      # IVJ_ECAPACITY
            return n.value, False
        _check(self.L, rc, "ivj_overlap_fused_rows_dev")
        return n.value, True

    def take_dev(self, src_ptr: int, elem_bytes: int, idx_ptr: int, n: int, dst_ptr: int, validity_ptr: int = 0):
        """ivj_take_dev: Arrow take of one 4- or 8-byte device column; negative indices -> 0 / null bit."""
        _check(self.L, self.L.ivj_take_dev(self.h, C.c_void_p(src_ptr), int(elem_bytes), C.c_void_p(idx_ptr), int(n),
                                            C.c_void_p(dst_ptr), C.c_void_p(validity_ptr or None)), "ivj_take_dev")

    def cluster_dev(self, ix: DeviceIndex, opts: _Opts, min_dist: int, cluster_ptr: int, start_ptr: int, end_ptr: int) -> int:
        n = C.c_int64(0)
        _check(self.L, self.L.ivj_cluster_dev(self.h, ix.handle, C.byref(opts), int(min_dist), C.c_void_p(cluster_ptr),
                                               C.c_void_p(start_ptr), C.c_void_p(end_ptr), C.byref(n)), "ivj_cluster_dev")
        return n.value

    def merge_dev(self, ix: DeviceIndex, opts: _Opts, min_dist: int, capacity: int, contig_ptr: int, start_ptr: int, end_ptr: int,
                  n_intervals_ptr: int):
        """-> (n_merged, fits)"""
        n = C.c_int64(0)
        rc = self.L.ivj_merge_dev(self.h, ix.handle, C.byref(opts), int(min_dist), int(capacity), C.c_void_p(contig_ptr or None),
                                  C.c_void_p(start_ptr or None), C.c_void_p(end_ptr or None), C.c_void_p(n_intervals_ptr or None), C.byref(n))
        if rc == -4:
            return n.value, False
        _check(self.L, rc, "ivj_merge_dev")
        return n.value, True

    def subtract_dev(self, right_ix: DeviceIndex, left: _Side, opts: _Opts, capacity: int, row_ptr: int, start_ptr: int, end_ptr: int):
        """-> (n_pieces, fits)"""
        n = C.c_int64(0)
        rc = self.L.ivj_subtract_dev(self.h, right_ix.handle, C.byref(left), C.byref(opts), int(capacity), C.c_void_p(row_ptr or None),
                                     C.c_void_p(start_ptr or None), C.c_void_p(end_ptr or None), C.byref(n))
        if rc == -4:
            return n.value, False
        _check(self.L, rc, "ivj_subtract_dev")
        return n.value, True

    def coverage_dev(self, ix: DeviceIndex, probe: _Side, opts: _Opts, coverage_ptr: int):
        _check(self.L, self.L.ivj_coverage_dev(self.h, ix.handle, C.byref(probe), C.byref(opts), C.c_void_p(coverage_ptr)),
               "ivj_coverage_dev")

    def count_overlaps_dev(self, ix: DeviceIndex, probe: _Side, opts: _Opts, counts_ptr: int):
        _check(self.L, self.L.ivj_count_overlaps_dev(self.h, ix.handle, C.byref(probe), C.byref(opts),
                                                      C.c_void_p(counts_ptr)), "ivj_count_overlaps_dev")

    def nearest_dev(self, ix: DeviceIndex, probe: _Side, opts: _Opts, idx_ptr: int, dist_ptr: int, nf_ptr: int):
        _check(self.L, self.L.ivj_nearest_dev(self.h, ix.handle, C.byref(probe), C.byref(opts), C.c_void_p(idx_ptr),
                                               C.c_void_p(dist_ptr), C.c_void_p(nf_ptr)), "ivj_nearest_dev")

    # ---- streaming: build side resident, probe side in bounded batches ---------
    def probe_stream(self, build, strict: bool, n_contigs: int, op: int = STREAM_OVERLAP, max_batch_rows: int = 8_000_000, k: int = 1,
                     include_overlaps: bool = True, partition_mode: int = 0, copy: bool = True) -> ProbeStream:
        """Open a streaming probe session (ivj_stream_open): see ProbeStream."""
        return ProbeStream(self, build, strict, n_contigs, op, max_batch_rows, k, include_overlaps, partition_mode, copy)

    def overlap_batches(self, probe, build, strict: bool, n_contigs: int, batch_rows: int = 8_000_000):
        """Generator of (probe_idx, build_idx) numpy batches over a probe side that is already in host arrays.

        The build side is sorted once and stays in HBM; the probe side goes through the device in batches of
        ``batch_rows`` rows with H2D / join / D2H of consecutive batches overlapped (ProbeStream), so device memory and
        the size of every result batch are bounded (the reference's streaming probe side + ``low_memory``:
        docs/developers.md:641-646, polars_bio/range_op.py:168).  probe_idx are rows of the WHOLE probe side."""
        pc, ps, pe = (_i32(a) for a in probe)
        n = pc.shape[0]
        if n == 0 or len(build[0]) == 0:
            return
        rows = int(min(batch_rows, n))
        with self.probe_stream(build, strict, n_contigs, STREAM_OVERLAP, rows, copy=False) as st:
            def emit(res):
                lo = res["batch"] * rows
                return (res["probe_idx"] + np.int32(lo)), res["build_idx"].copy()
            for lo in range(0, n, rows):
                hi = min(lo + rows, n)
                res = st.submit((pc[lo:hi], ps[lo:hi], pe[lo:hi]))
                if res is not None:
                    yield emit(res)
            while True:
                res = st.flush()
                if res is None:
                    break
                yield emit(res)

    # ---- raw device memory (callers without torch) --------------------------
    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _check(self.L, self.L.ivj_dev_alloc(self.h, nbytes, C.byref(p)), "ivj_dev_alloc")
        return p.value or 0

    def dev_free(self, ptr: int):
        _check(self.L, self.L.ivj_dev_free(self.h, C.c_void_p(ptr)), "ivj_dev_free")

    def h2d(self, dst_ptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        _check(self.L, self.L.ivj_memcpy_h2d(self.h, C.c_void_p(dst_ptr), arr.ctypes.data, arr.nbytes), "ivj_memcpy_h2d")

    def d2h(self, arr: np.ndarray, src_ptr: int):
        assert arr.flags.c_contiguous
        _check(self.L, self.L.ivj_memcpy_d2h(self.h, arr.ctypes.data, C.c_void_p(src_ptr), arr.nbytes), "ivj_memcpy_d2h")


UNIQUE_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library: rank 0 makes it, every other rank receives the 128 bytes by whatever channel
    the host has (a file, MPI, a torch.distributed object broadcast ...)."""
    L = load_library()
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    _check(L, L.ivj_comm_unique_id(buf), "ivj_comm_unique_id")
    return buf.raw


class Comm:
    """RCCL communicator of the library (include/ivjoin.h, ivj_comm_*): one rank per GPU; the all-gatherv of result batches
    and the sharded overlap with the exchange overlapping the join run inside libivjoin_hip.so, no PyTorch on the data path."""

    def __init__(self, engine: Engine, unique_id: Optional[bytes], rank: int, world: int, _handle=None):
        self.engine, self.rank, self.world = engine, rank, world
        self.L = load_library()
        if _handle is not None:
            self.h = _handle
            return
        h = C.c_void_p()
        idbuf = C.create_string_buffer(unique_id, UNIQUE_ID_BYTES) if unique_id is not None else None
        _check(self.L, self.L.ivj_comm_create(engine.h, idbuf, int(rank), int(world), C.byref(h)), "ivj_comm_create")
        self.h = h

    @staticmethod
    def create_local(engines):
        """One process, one Engine per device: ncclCommInitAll.  Collective calls on the returned communicators must come
        from one host thread per rank."""
        L = load_library()
        n = len(engines)
        ctxs = (C.c_void_p * n)(*[e.h for e in engines])
        out = (C.c_void_p * n)()
        _check(L, L.ivj_comm_create_local(ctxs, n, out), "ivj_comm_create_local")
        return [Comm(e, None, i, n, _handle=C.c_void_p(out[i])) for i, e in enumerate(engines)]

    def close(self):
        if getattr(self, "h", None):
            self.L.ivj_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def allgather_counts(self, n_local: int):
        counts = (C.c_int64 * self.world)()
        _check(self.L, self.L.ivj_allgather_counts(self.h, int(n_local), counts), "ivj_allgather_counts")
        return [int(x) for x in counts]

    def allgatherv_dev(self, send_ptrs, recv_ptrs, elem_bytes: int, counts):
        n = len(send_ptrs)
        sp = (C.c_void_p * n)(*[C.c_void_p(p) for p in send_ptrs])
        rp = (C.c_void_p * n)(*[C.c_void_p(p) for p in recv_ptrs])
        cn = (C.c_int64 * self.world)(*[int(x) for x in counts])
        with self.engine.lock:
            _check(self.L, self.L.ivj_allgatherv_dev(self.h, sp, rp, n, int(elem_bytes), cn), "ivj_allgatherv_dev")

    def overlap_allgather_dev(self, ix: "DeviceIndex", probe: _Side, opts: _Opts, n_chunks: int, probe_idx_ptr: int, build_idx_ptr: int,
                              capacity: int):
        """-> (n_total, n_local, fits): this rank's shard joined and all-gathered, the exchange overlapping the join."""
        nt, nl = C.c_int64(0), C.c_int64(0)
        with self.engine.lock:
            rc = self.L.ivj_overlap_allgather_dev(self.h, ix.handle, C.byref(probe), C.byref(opts), int(n_chunks), C.c_void_p(probe_idx_ptr),
                                                  C.c_void_p(build_idx_ptr), int(capacity), C.byref(nt), C.byref(nl))
        if rc == -4:
            return nt.value, nl.value, False
        _check(self.L, rc, "ivj_overlap_allgather_dev")
        return nt.value, nl.value, True

    def count_overlaps_allgather_dev(self, ix: "DeviceIndex", probe: _Side, opts: _Opts, n_total: int, counts_ptr: int):
        """count_overlaps of this rank's shard (probe.row_id = global rows) + the exchange of the per-probe results: the int64
        column counts_ptr[n_total] holds every probe row's count, in global probe order, on every rank."""
        with self.engine.lock:
            _check(self.L, self.L.ivj_count_overlaps_allgather_dev(self.h, ix.handle, C.byref(probe), C.byref(opts), int(n_total), C.c_void_p(counts_ptr)),
                   "ivj_count_overlaps_allgather_dev")

    def nearest_allgather_dev(self, ix: "DeviceIndex", probe: _Side, opts: _Opts, n_total: int, idx_ptr: int, dist_ptr: int, nf_ptr: int):
        """nearest of this rank's shard + the exchange: idx_ptr[n_total * k] (int32), dist_ptr[n_total * k] (int64),
        nf_ptr[n_total] (int32) in global probe order on every rank."""
        with self.engine.lock:
            _check(self.L, self.L.ivj_nearest_allgather_dev(self.h, ix.handle, C.byref(probe), C.byref(opts), int(n_total), C.c_void_p(idx_ptr),
                                                            C.c_void_p(dist_ptr), C.c_void_p(nf_ptr)), "ivj_nearest_allgather_dev")


# ---- the one-call Arrow entry (include/ivjoin.h: ivj_*_arrow_stream): what a C / Rust host binds, driven from Python -------------
def _export_stream(src):
    """Anything Arrow-streamable (pyarrow.Table / RecordBatchReader / an object with __arrow_c_stream__) -> a filled
    struct ArrowArrayStream (the library drains and releases it)."""
    import pyarrow as pa
    if isinstance(src, pa.Table):
        src = src.to_reader()
    elif not isinstance(src, pa.RecordBatchReader):
        src = pa.RecordBatchReader.from_stream(src)
    s = _ArrowStream()
    src._export_to_c(C.addressof(s))
    return s


def _import_stream(s: _ArrowStream):
    import pyarrow as pa
    return pa.RecordBatchReader._import_from_c(C.addressof(s))


def _names3(cols):
    if cols is None:
        return None
    return (C.c_char_p * 3)(*[str(c).encode() for c in cols])


def arrow_encode_keys(df1, df2, cols1=None, cols2=None):
    """ivj_arrow_encode_keys: -> ((contig, start, end) of df1, of df2, dictionary names); host only."""
    L = load_library()
    s1, s2 = _export_stream(df1), _export_stream(df2)
    k = _ArrowKeys()
    _check(L, L.ivj_arrow_encode_keys(C.addressof(s1), C.addressof(s2), _names3(cols1), _names3(cols2), C.byref(k)), "ivj_arrow_encode_keys")
    try:
        col = lambda p, n: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(n,)).copy() if n else np.empty(0, np.int32)
        side1 = (col(k.contig1, k.n1), col(k.start1, k.n1), col(k.end1, k.n1))
        side2 = (col(k.contig2, k.n2), col(k.start2, k.n2), col(k.end2, k.n2))
        offs = np.ctypeslib.as_array(C.cast(k.name_offsets, C.POINTER(C.c_int64)), shape=(k.n_contigs + 1,))
        raw = C.string_at(k.name_bytes, int(offs[-1]))
        names = [raw[offs[i]:offs[i + 1]].decode() for i in range(k.n_contigs)]
        return side1, side2, names
    finally:
        L.ivj_arrow_keys_free(C.byref(k))


def arrow_take_stream(src, idx, batch_rows: int = 0):
    """ivj_arrow_take_stream: rows idx (negative: a null row) of an Arrow stream as a new pyarrow.RecordBatchReader; host only."""
    L = load_library()
    s, out = _export_stream(src), _ArrowStream()
    idx = np.ascontiguousarray(idx, np.int64)
    _check(L, L.ivj_arrow_take_stream(C.addressof(s), C.c_void_p(idx.ctypes.data), len(idx), int(batch_rows), C.addressof(out)), "ivj_arrow_take_stream")
    return _import_stream(out)


def _arrow_stream_call(engine, op: str, df1, df2, cols1, cols2, strict: bool, suffixes, k: int = 1, include_overlaps: bool = True,
                       distance: bool = True, batch_rows: int = 0, limit=None, lazy: bool = False, max_batch_rows: int = 0):
    L = load_library()
    s1, s2, out = _export_stream(df1), _export_stream(df2), _ArrowStream()
    opts = make_opts(strict, 0, k, include_overlaps)
    lim = -1 if limit is None else int(limit)
    sfx = [None if x is None else str(x).encode() for x in suffixes]
    a = (engine.h, C.addressof(s1), C.addressof(s2), _names3(cols1), _names3(cols2), C.byref(opts))
    with engine.lock:
        if op == "overlap":
            rc = (L.ivj_overlap_arrow_stream_lazy(*a, sfx[0], sfx[1], int(batch_rows), int(max_batch_rows), lim, C.addressof(out)) if lazy else
                  L.ivj_overlap_arrow_stream(*a, sfx[0], sfx[1], int(batch_rows), lim, C.addressof(out)))
        elif op == "count_overlaps":
            rc = (L.ivj_count_overlaps_arrow_stream_lazy(*a, sfx[0], int(batch_rows), int(max_batch_rows), lim, C.addressof(out)) if lazy else
                  L.ivj_count_overlaps_arrow_stream(*a, sfx[0], int(batch_rows), lim, C.addressof(out)))
        else:
            rc = (L.ivj_nearest_arrow_stream_lazy(*a, sfx[0], sfx[1], 1 if distance else 0, int(batch_rows), int(max_batch_rows), lim, C.addressof(out)) if lazy else
                  L.ivj_nearest_arrow_stream(*a, sfx[0], sfx[1], 1 if distance else 0, int(batch_rows), lim, C.addressof(out)))
    _check(L, rc, f"ivj_{op}_arrow_stream" + ("_lazy" if lazy else ""))
    reader = _import_stream(out)
    if not lazy:
        return reader
    # the lazy stream works on the engine's context whenever a batch is pulled: every pull is made under the engine's lock
    import pyarrow as pa

    def pull():
        while True:
            with engine.lock:
                try:
                    b = reader.read_next_batch()
                except StopIteration:
                    return
            yield b
    return pa.RecordBatchReader.from_batches(reader.schema, pull())


def overlap_arrow_stream(engine, df1, df2, strict: bool, cols1=None, cols2=None, suffixes=("_1", "_2"), batch_rows: int = 0, limit=None,
                         lazy: bool = False, max_batch_rows: int = 0):
    """ivj_overlap_arrow_stream[_lazy] -> pyarrow.RecordBatchReader of the joined rows (every column of both sides, suffixed).
    lazy: df1 is pulled batch by batch while the result is read (host memory bounded by max_batch_rows, not by len(df1))."""
    return _arrow_stream_call(engine, "overlap", df1, df2, cols1, cols2, strict, suffixes, batch_rows=batch_rows, limit=limit, lazy=lazy, max_batch_rows=max_batch_rows)


def count_overlaps_arrow_stream(engine, df1, df2, strict: bool, cols1=None, cols2=None, suffix="", batch_rows: int = 0, limit=None,
                                lazy: bool = False, max_batch_rows: int = 0):
    return _arrow_stream_call(engine, "count_overlaps", df1, df2, cols1, cols2, strict, (suffix, None), batch_rows=batch_rows, limit=limit, lazy=lazy,
                              max_batch_rows=max_batch_rows)


def nearest_arrow_stream(engine, df1, df2, strict: bool, cols1=None, cols2=None, suffixes=("_1", "_2"), k: int = 1, include_overlaps: bool = True,
                         distance: bool = True, batch_rows: int = 0, limit=None, lazy: bool = False, max_batch_rows: int = 0):
    return _arrow_stream_call(engine, "nearest", df1, df2, cols1, cols2, strict, suffixes, k, include_overlaps, distance, batch_rows, limit, lazy, max_batch_rows)


_default_engine: Optional[Engine] = None
_default_lock = threading.Lock()


def default_engine() -> Engine:
    """Process-wide engine (its native calls are serialised by Engine.lock; use one Engine per thread for
    concurrent joins).  One device: the ``ivj.device`` option when it was set explicitly, else LOCAL_RANK (one
    process per GPU under torch.distributed.run), else 0.  Several devices (``ivj.devices`` = "0,1,..", or
    ``ivj.num_gpus`` > 1 on a host with that many GPUs): a
    ``multi.MultiEngine`` -- one context and one host thread per device, contigs dealt to the devices."""
    global _default_engine
    with _default_lock:
        if _default_engine is None:
            from .multi import MultiEngine, requested_devices
            devs = requested_devices()
            _default_engine = MultiEngine(devs) if len(devs) > 1 else Engine(devs[0])
        return _default_engine


def reset_default_engine():
    """Drop the process-wide engine (the next call creates a new one, e.g. after ``ivj.device`` changed).  The old engine is
    NOT closed here: another thread or an unconsumed lazy reader may still be running on it -- its context goes when the last
    reference does (Engine.__del__)."""
    global _default_engine
    with _default_lock:
        _default_engine = None
