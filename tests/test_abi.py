"""CPU checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/ivjoin.h declares; the product path has no CPU fallback and never touches
the oracle."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "polars-bio_amd", "polars_bio_amd")


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "ivjoin.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ivj_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from polars_bio_amd import _engine
    L = _engine.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 20
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_engine.ABI_SYMBOLS) == declared
    assert b"gfx950" in L.ivj_version()


def test_library_contains_gfx950_code_object():
    """The fat binary section must carry a gfx950 code object (and nothing else)."""
    from polars_bio_amd import _engine
    blob = open(_engine.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"hipv4-amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_no_cpu_fallback_without_device():
    """Without a GPU the engine must fail loudly, not compute on the CPU."""
    from polars_bio_amd import _engine
    if _engine.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_engine.EngineError):
        _engine.Engine(0)
    import pandas as pd
    import polars_bio_amd as pb
    df = pd.DataFrame({"chrom": ["chr1"], "start": [1], "end": [5]})
    df.attrs["coordinate_system_zero_based"] = True
    with pytest.raises(_engine.EngineError):
        pb.overlap(df, df, output_type="pandas.DataFrame")


def test_product_path_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "polars-bio_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", "").lower() or f in ("_engine.py", "__init__.py"), (dirpath, f)
    # the two files allowed above only mention the word in prose; make sure they do not import it
    for f in ("_engine.py", "__init__.py"):
        src = open(os.path.join(PKG, f)).read()
        assert not re.search(r"^\s*(from|import)\s+.*oracle", src, flags=re.M)


def test_error_codes_for_bad_arguments():
    from polars_bio_amd import _engine
    import ctypes as C
    L = _engine.load_library()
    assert L.ivj_device_count(None) == -1
    assert b"NULL" in L.ivj_last_error()
    assert L.ivj_ctx_sync(None) == -1


def test_arrow_c_data_import_of_a_side_is_zero_copy():
    """ivj_side_from_arrow: the three key columns of a record batch are viewed in place (offsets of sliced batches
    applied), extra columns ignored, wrong types / nulls / missing columns refused.  Needs no device."""
    import ctypes as C
    import numpy as np
    import pyarrow as pa
    from polars_bio_amd import _engine
    n = 1000
    rng = np.random.default_rng(1)
    cols = {"score": pa.array(rng.random(n)), "end": pa.array(rng.integers(0, 1 << 30, n).astype(np.int32)),
            "contig": pa.array(rng.integers(0, 24, n).astype(np.int32)), "start": pa.array(rng.integers(0, 1 << 30, n).astype(np.int32))}
    batch = pa.record_batch(cols)
    for b in (batch, batch.slice(17, 400)):
        side, keep = _engine.side_from_arrow(b)
        assert side.n == b.num_rows
        for name in ("contig", "start", "end"):
            got = np.ctypeslib.as_array(C.cast(getattr(side, name), C.POINTER(C.c_int32)), shape=(b.num_rows,))
            exp = b.column(name).to_numpy()
            assert (got == exp).all()
            assert got.ctypes.data == exp.ctypes.data          # same memory: nothing was copied
        del keep
    bad = [pa.record_batch({"contig": cols["contig"], "start": cols["start"]}),                                   # no end
           pa.record_batch({"contig": cols["contig"], "start": cols["start"], "end": pa.array(np.arange(n, dtype=np.int64))}),
           pa.record_batch({"contig": cols["contig"], "start": cols["start"], "end": pa.array([None] + [1] * (n - 1), type=pa.int32())})]
    for b in bad:
        with pytest.raises(_engine.EngineError):
            _engine.side_from_arrow(b)
    empty, _ = _engine.side_from_arrow(batch.slice(0, 0))
    assert empty.n == 0


def test_host_memory_probe_reads_memavailable(monkeypatch):
    """The host entry points gate big results on MemAvailable (reclaimable page cache counts), not on MemFree; the
    probe can be pinned through IVJ_HOST_MEM_AVAILABLE."""
    from polars_bio_amd import _engine
    L = _engine.load_library()
    info = dict(l.split(":", 1) for l in open("/proc/meminfo").read().splitlines() if ":" in l)
    avail = int(info["MemAvailable"].split()[0]) * 1024
    free = int(info["MemFree"].split()[0]) * 1024
    got = L.ivj_host_mem_available()
    assert abs(got - avail) <= max(avail // 4, 1 << 30), (got, avail, free)
    monkeypatch.setenv("IVJ_HOST_MEM_AVAILABLE", str(12345 << 20))
    assert L.ivj_host_mem_available() == 12345 << 20
