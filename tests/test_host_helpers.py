"""Native host passes of the front door (include/ivjoin.h "host-side helpers", csrc/host_frontdoor.hip.h) against numpy / pyarrow:
no device work, so they run in the CPU suite.  Reference counterparts: the int32 coordinate limit
(/root/reference/docs/features/operations.md:36-37), the chrom join key as exact string equality, the column
gathers of the renaming SELECT (/root/reference/src/operation.rs:272-301)."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from polars_bio_amd import _arrow as A
from polars_bio_amd import _host as H


@pytest.mark.parametrize("dtype", [np.int64, np.uint64, np.int32, np.uint32, np.int16, np.uint16, np.int8, np.uint8])
def test_narrow_i32_matches_numpy_for_every_integer_width(dtype):
    rng = np.random.default_rng(1)
    info = np.iinfo(dtype)
    lo, hi = max(info.min, -(1 << 31)), min(info.max, (1 << 31) - 1)
    for n in (0, 1, 63, 64, 65, 300_001):
        a = rng.integers(lo, hi, n, dtype=np.int64, endpoint=True).astype(dtype)
        out, mn, mx = H.narrow_i32(a)
        assert out.dtype == np.int32 and (out == a.astype(np.int32)).all()
        if n:
            assert mn == int(a.min()) and mx == int(a.max())


def test_narrow_reports_values_beyond_int32_and_the_front_door_refuses_them():
    a = np.array([5, (1 << 31), 7], np.int64)
    _, mn, mx = H.narrow_i32(a)
    assert mn == 5 and mx == 1 << 31
    _, mn, mx = H.narrow_i32(np.array([1, 2 ** 63 + 5], np.uint64))          # beyond int64: saturates, still refused
    assert mx == 2 ** 63 - 1
    with pytest.raises(ValueError, match="does not fit int32"):
        A._coord_to_i32(pa.chunked_array([pa.array(a)]), "start")
    with pytest.raises(ValueError, match="does not fit int32"):
        A._coord_to_i32(pa.chunked_array([pa.array(np.array([-(1 << 31) - 1], np.int64))]), "start")
    ok = A._coord_to_i32(pa.chunked_array([pa.array(np.array([-(1 << 31), (1 << 31) - 1], np.int64))]), "start")
    assert ok.tolist() == [-(1 << 31), (1 << 31) - 1]


def _check_encode(arr: pa.Array):
    out = np.empty(len(arr), np.int32)
    offs, data, valid, bit0 = A._string_buffers(arr)
    rows = H.encode_utf8(offs, data, valid, bit0, len(arr), out)
    assert rows is not None
    values = arr.take(pa.array(rows)).to_pylist()
    expect = arr.to_pylist()
    assert len(set(values)) == len(values) and None not in values          # distinct, and every one occurs
    assert [values[i] if i >= 0 else None for i in out.tolist()] == expect
    first = {}
    for i, v in enumerate(expect):
        if v is not None:
            first.setdefault(v, i)
    assert values == sorted(first, key=first.get)                          # first-occurrence order
    return values


@pytest.mark.parametrize("typ", [pa.string(), pa.large_string()])
def test_encode_utf8_values_nulls_empty_strings_long_names_and_slices(typ):
    rng = np.random.default_rng(2)
    names = ["chr1", "chr2", "chr10", "chrX", "", "chrUn_KI270302v1", "chrUn_KI270303v1", "chrUn_KI270302v2", "12345678", "123456789", "1234567",
             "HLA-DRB1*15:01:01:01", "HLA-DRB1*15:01:01:02"]
    for n in (1, 7, 70_000, 400_003):
        vals = [names[i] for i in rng.integers(0, len(names), n)]
        for k in rng.integers(0, n, max(1, n // 50)):
            vals[k] = None
        arr = pa.array(vals, type=typ)
        _check_encode(arr)
        if n > 10:
            _check_encode(arr.slice(3, n - 7))                                 # offsets and validity bits that do not start at 0
    _check_encode(pa.array(["only"] * 100_000, type=typ))
    assert _check_encode(pa.array([""] * 5, type=typ)) == [""]
    empty = pa.array([], type=typ)
    out = np.empty(0, np.int32)
    offs, data, valid, bit0 = A._string_buffers(empty)
    assert len(H.encode_utf8(offs, data, valid, bit0, 0, out)) == 0


def test_encode_utf8_hands_columns_with_thousands_of_values_back_to_the_caller():
    vals = [f"scaffold_{i}" for i in range(H.MAX_DICT + 50)] * 3
    arr = pa.array(vals)
    out = np.empty(len(arr), np.int32)
    offs, data, valid, bit0 = A._string_buffers(arr)
    assert H.encode_utf8(offs, data, valid, bit0, len(arr), out) is None
    d, ids = A._encode_chrom(pa.chunked_array([arr]))                          # the front door falls back to pyarrow's encoder
    assert len(d) == H.MAX_DICT + 50 and [d[i].as_py() for i in ids[:10].tolist()] == vals[:10]


def test_encode_chrom_over_chunks_of_mixed_kinds_shares_one_dictionary():
    a = pa.array(["chr2", "chr1", None, "chr2"])
    b = pa.array(["chr3", "chr1"], type=pa.large_string())
    c = pa.array(["chrX", "chr1", "chr2"]).dictionary_encode()
    c = pa.DictionaryArray.from_arrays(pa.array([0, 0, 2], pa.int8()), pa.array(["chr9", "unused", "chr1"]))
    col = pa.chunked_array([a.cast(pa.large_string()), b])
    d, ids = A._encode_chrom(col)
    got = [d[i].as_py() if i >= 0 else None for i in ids.tolist()]
    assert got == ["chr2", "chr1", None, "chr2", "chr3", "chr1"]
    d, ids = A._encode_chrom(pa.chunked_array([c]))
    assert sorted(d.to_pylist()) == ["chr1", "chr9"]                           # the entry no row refers to is dropped
    assert [d[i].as_py() for i in ids.tolist()] == ["chr9", "chr9", "chr1"]


@pytest.mark.parametrize("idx_dtype", [np.int8, np.int16, np.int32, np.int64])
def test_remap_i32_and_seen(idx_dtype):
    rng = np.random.default_rng(3)
    table = np.array([7, -1, 3, 0, 9], np.int32)
    idx = rng.integers(0, 4, 200_001).astype(idx_dtype)                        # slot 4 never used
    idx[::17] = -1
    out = np.empty(len(idx), np.int32)
    seen = np.zeros(len(table), np.uint8)
    H.remap_i32(idx, table, out, seen)
    assert (out == np.where(idx < 0, -1, table[np.maximum(idx, 0)])).all()
    assert seen.tolist() == [1, 1, 1, 1, 0]
    with pytest.raises(Exception, match="outside the dictionary"):
        H.remap_i32(np.array([0, 5], idx_dtype), table, np.empty(2, np.int32))


@pytest.mark.parametrize("dtype", [np.int64, np.float64, np.int32, np.float32, np.uint32])
def test_take_matches_numpy_and_negative_indices_yield_zero(dtype):
    rng = np.random.default_rng(4)
    src = (rng.random(100_003) * 1e6).astype(dtype)
    idx = rng.integers(0, len(src), 250_001).astype(np.int32)
    assert (H.take(src, idx) == src[idx]).all()
    idx[::9] = -1
    got = H.take(src, idx)
    assert (got[idx >= 0] == src[idx[idx >= 0]]).all() and (got[idx < 0] == 0).all()
    assert len(H.take(src, np.empty(0, np.int32))) == 0


def test_take_rows_native_and_pyarrow_columns_agree_with_arrow_take():
    rng = np.random.default_rng(5)
    n = 50_000
    t = pa.table({"chrom": pa.array(rng.choice(["a", "bb", "ccc"], n)), "x": rng.integers(0, 1 << 40, n), "y": rng.random(n).astype(np.float32),
                  "s": pa.array([f"v{i}" for i in range(n)]), "z": pa.array([None if i % 11 == 0 else i for i in range(n)], pa.int64())})
    idx = rng.integers(0, n, 300_000).astype(np.int32)
    got = A.take_rows(t, idx)
    assert got.equals(t.take(pa.array(idx)))
    idx[::13] = -1
    got = A.take_rows(t, idx, nullable=True)
    exp = t.take(pa.array(np.where(idx < 0, 0, idx), mask=idx < 0))
    assert got.equals(exp)
    d, ids = A._encode_chrom(t.column("chrom"))
    with_dict = A.take_rows(t, idx, nullable=True, chrom=("chrom", ids, d))
    assert with_dict.column("chrom").to_pylist() == exp.column("chrom").to_pylist()


def test_widen_i64():
    a = np.array([-(1 << 31), -1, 0, (1 << 31) - 1] * 70_000, np.int32)
    assert (H.widen_i64(a) == a.astype(np.int64)).all()


def test_host_worker_pool_under_concurrent_callers_and_after_fork():
    """The native passes share one process-wide worker pool (csrc/host_mem.hip.h: host_parallel): a caller that finds it busy runs on
    threads of its own, a forked child starts a fresh pool.  Six Python threads hammer two passes; a forked child runs one."""
    import os
    import threading
    rng = np.random.default_rng(0)
    a = rng.integers(-2 ** 31, 2 ** 31 - 1, 1_500_003, dtype=np.int64)
    src = rng.integers(0, 2 ** 60, 500_003, dtype=np.int64)
    idx = rng.integers(0, len(src), 1_000_001).astype(np.int32)
    exp_n, exp_t = a.astype(np.int32), src[idx]
    errs = []

    def worker(k):
        for it in range(15):
            out, mn, mx = H.narrow_i32(a)
            if not (out == exp_n).all() or mn != a.min() or mx != a.max():
                errs.append(("narrow", k, it))
            if not (H.take(src, idx) == exp_t).all():
                errs.append(("take", k, it))
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs[:3]
    pid = os.fork()
    if pid == 0:
        ok = (H.narrow_i32(a)[0] == exp_n).all()
        os._exit(0 if ok else 3)
    _, st = os.waitpid(pid, 0)
    assert os.WEXITSTATUS(st) == 0


def test_host_scatter_is_the_mirror_of_take():
    """ivj_host_scatter: dst[idx[i]] = src[i] for 4- / 8-byte values and k-wide rows, with the optional local -> global remap of
    int32 values (negative stays -1); an index outside the destination is refused before anything is written (round 5:
    MultiEngine puts per-probe results back with it instead of `out[pid] = c` under the GIL)."""
    from polars_bio_amd import _host as H, _engine as E
    rng = np.random.default_rng(1)
    n, m = 200_000, 300_000
    pid = rng.permutation(m)[:n].astype(np.int32)
    out = np.zeros(m, np.int64)
    c = rng.integers(0, 100, n)
    H.scatter(out, pid, c)
    ref = np.zeros(m, np.int64); ref[pid] = c
    assert (out == ref).all()
    f = np.zeros(m, np.int32)
    H.scatter(f, pid, (c % 2).astype(np.int32))
    assert (f[pid] == c % 2).all() and f.sum() == (c % 2).sum()
    idx = np.full((m, 3), -7, np.int32)
    i = rng.integers(-1, 50, (n, 3)).astype(np.int32)
    bid = (np.arange(50) * 3).astype(np.int32)
    H.scatter(idx, pid, i, remap=bid)
    ref = np.full((m, 3), -7, np.int32); ref[pid] = np.where(i >= 0, bid[np.where(i >= 0, i, 0)], -1)
    assert (idx == ref).all()
    d = np.full((m, 3), -1, np.int64)
    dv = rng.integers(0, 10**12, (n, 3))
    H.scatter(d, pid, dv)
    assert (d[pid] == dv).all()
    before = out.copy()
    with pytest.raises(E.EngineError, match="outside the destination"):
        H.scatter(out, np.array([5, m], np.int32), np.array([1, 2]))
    assert (out == before).all()


def test_datafusion_dataframe_output_registers_the_arrow_result(monkeypatch):
    """output_type='datafusion.DataFrame' (reference: range_op_helpers.py:362-370): the Arrow result goes through
    SessionContext().from_arrow when `datafusion` imports, and is an ImportError naming the package when it does not."""
    import sys
    import types
    import pyarrow as pa
    import pytest
    from polars_bio_amd import _arrow
    t = pa.table({"contig_1": ["chr1"], "pos_start_1": [1], "pos_end_1": [5]})
    monkeypatch.setitem(sys.modules, "datafusion", None)                       # import datafusion -> ImportError
    with pytest.raises(ImportError, match="datafusion"):
        _arrow.from_arrow(t, "datafusion.DataFrame", True)
    seen = {}

    class _Ctx:
        def from_arrow(self, table):
            seen["table"] = table
            return ("df", table.num_rows)

    monkeypatch.setitem(sys.modules, "datafusion", types.SimpleNamespace(SessionContext=_Ctx))
    assert _arrow.from_arrow(t, "datafusion.DataFrame", True) == ("df", 1)
    assert seen["table"].column_names == t.column_names
    assert seen["table"].schema.metadata                                       # the coordinate-system metadata rides along
