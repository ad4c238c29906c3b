"""BASELINE.json configs[0]: pb.overlap() on two 1k-row synthetic (chrom, start, end) frames, 1 contig.

The reference runs this configuration on its Rust / COITrees CPU path (polars_bio/range_op.py:117-256); here the named
workload ``overlap_1k_1k_1contig`` goes through the same front door (string chroms; pandas and pyarrow frames) and through
the C-ABI host entry point ``ivj_overlap``, and both are compared pair for pair with the brute-force oracle.
"""
import numpy as np
import pandas as pd
import pyarrow as pa
import pytest

import polars_bio_amd as pb
from polars_bio_amd import range_op, synth
from _util import OracleEngine
from oracle import oracle as O

WORKLOAD = "overlap_1k_1k_1contig"


@pytest.fixture(params=["cpu", pytest.param("gpu", marks=pytest.mark.gpu)])
def engine(request, monkeypatch):
    if request.param == "cpu":
        monkeypatch.setattr(range_op, "default_engine", lambda: OracleEngine())
    return request.param


def _expected():
    probe, build, nc = synth.workload(WORKLOAD)
    ep, eb = O.overlap_brute(O.Side(*probe), O.Side(*build), True)
    return probe, build, nc, ep, eb


def _frames(probe, build, kind):
    def one(side):
        c, s, e = side
        d = {"chrom": np.array(synth.CONTIG_NAMES, dtype=object)[c], "start": s.astype(np.int64), "end": e.astype(np.int64)}
        if kind == "pandas":
            df = pd.DataFrame(d)
            df.attrs["coordinate_system_zero_based"] = True
            return df
        t = pa.table({"chrom": pa.array(d["chrom"], pa.string()), "start": pa.array(d["start"]), "end": pa.array(d["end"])})
        return t.replace_schema_metadata({"coordinate_system_zero_based": "true"})
    return one(probe), one(build)


@pytest.mark.parametrize("kind", ["pandas", "pyarrow"])
def test_config1_front_door(engine, kind):
    probe, build, nc, ep, eb = _expected()
    assert nc == 1 and len(probe[0]) == 1000 and len(build[0]) == 1000
    df1, df2 = _frames(probe, build, kind)
    out = "pandas.DataFrame" if kind == "pandas" else "pyarrow.Table"
    res = pb.overlap(df1, df2, output_type=out)
    res = res if kind == "pandas" else res.to_pandas()
    assert len(res) == len(ep) > 0
    exp = pd.DataFrame({"chrom_1": np.array(synth.CONTIG_NAMES, dtype=object)[probe[0][ep]],
                        "start_1": probe[1][ep].astype(np.int64), "end_1": probe[2][ep].astype(np.int64),
                        "chrom_2": np.array(synth.CONTIG_NAMES, dtype=object)[build[0][eb]],
                        "start_2": build[1][eb].astype(np.int64), "end_2": build[2][eb].astype(np.int64)})
    key = list(exp.columns)
    got = res[key].astype({"chrom_1": object, "chrom_2": object}).sort_values(key).reset_index(drop=True)
    pd.testing.assert_frame_equal(got, exp.sort_values(key).reset_index(drop=True), check_dtype=False)


@pytest.mark.gpu
def test_config1_c_abi_host_entry():
    """ivj_overlap (the function the reference's FFI would bind, include/ivjoin.h) on the named workload, every partition policy."""
    from polars_bio_amd import _engine
    probe, build, nc, ep, eb = _expected()
    eng = _engine.Engine(0)
    for pm in (0, 1, 2, 6):
        p, b = eng.overlap(probe, build, True, nc, partition_mode=pm)
        o = np.argsort(p, kind="stable")
        assert len(p) == len(ep) and (p[o] == ep).all() and (b[o] == eb).all(), pm
