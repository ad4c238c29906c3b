"""Coordinate-system metadata -> FilterOp (reference: polars_bio/_metadata.py:267-362,
polars_bio/range_op.py:56-84).  Only the bool -> Strict/Weak mapping matters to
the hot path; this module keeps the reference's error/warning behaviour.

Carriers: pandas ``df.attrs``, pyarrow schema metadata, polars ``config_meta``
(when polars + polars-config-meta are installed).
"""
from __future__ import annotations

import warnings
from typing import Optional

import pyarrow as pa

from .constants import (COORDINATE_SYSTEM_KEY, POLARS_BIO_COORDINATE_SYSTEM_CHECK,
                        POLARS_BIO_COORDINATE_SYSTEM_ZERO_BASED)
from .context import get_option
from .exceptions import CoordinateSystemMismatchError, MissingCoordinateSystemError

try:  # optional front doors
    import pandas as pd
except ImportError:  # pragma: no cover
    pd = None
try:
    import polars as pl
except ImportError:
    pl = None


def _parse_bool(v) -> Optional[bool]:
    if v is None:
        return None
    if isinstance(v, bool):
        return v
    if isinstance(v, bytes):
        v = v.decode()
    if isinstance(v, str):
        return v.strip().lower() in ("true", "1")
    return bool(v)


def get_coordinate_system(df) -> Optional[bool]:
    """True = 0-based half-open, False = 1-based closed, None = no metadata."""
    if pd is not None and isinstance(df, pd.DataFrame):
        return _parse_bool(df.attrs.get(COORDINATE_SYSTEM_KEY))
    if isinstance(df, (pa.Table, pa.RecordBatch, pa.RecordBatchReader)):
        md = df.schema.metadata or {}
        return _parse_bool(md.get(COORDINATE_SYSTEM_KEY.encode()))
    if pl is not None and isinstance(df, (pl.DataFrame, pl.LazyFrame)):
        meta = getattr(df, "config_meta", None)
        if meta is not None:
            try:
                return _parse_bool(meta.get_metadata().get(COORDINATE_SYSTEM_KEY))
            except Exception:
                return None
        return None
    if isinstance(df, str):
        if df.endswith(".parquet"):
            import pyarrow.parquet as pq
            try:
                md = pq.read_schema(df).metadata or {}
                return _parse_bool(md.get(COORDINATE_SYSTEM_KEY.encode()))
            except Exception:
                return None
        return None
    return None


def set_coordinate_system(df, zero_based: bool):
    """Attach the metadata; returns the (possibly new) object."""
    if pd is not None and isinstance(df, pd.DataFrame):
        df.attrs[COORDINATE_SYSTEM_KEY] = bool(zero_based)
        return df
    if isinstance(df, pa.Table):
        md = dict(df.schema.metadata or {})
        md[COORDINATE_SYSTEM_KEY.encode()] = b"true" if zero_based else b"false"
        return df.replace_schema_metadata(md)
    if pl is not None and isinstance(df, (pl.DataFrame, pl.LazyFrame)):
        meta = getattr(df, "config_meta", None)
        if meta is not None:
            meta.set(**{COORDINATE_SYSTEM_KEY: bool(zero_based)})
        return df
    return df


def _type_name(df) -> str:
    return type(df).__module__.split(".")[0] + "." + type(df).__name__ if not isinstance(df, str) else f"path '{df}'"


def validate_coordinate_systems(df1, df2) -> bool:
    """Same contract as the reference's validate_coordinate_systems
    (polars_bio/_metadata.py:267-362): returns zero_based."""
    cs1, cs2 = get_coordinate_system(df1), get_coordinate_system(df2)
    check = (get_option(POLARS_BIO_COORDINATE_SYSTEM_CHECK) or "false").lower() == "true"
    if cs1 is None or cs2 is None:
        if check:
            which = df1 if cs1 is None else df2
            raise MissingCoordinateSystemError(
                f"{_type_name(which)} is missing coordinate system metadata.\n\n"
                f"Set df.attrs['{COORDINATE_SYSTEM_KEY}'] (pandas), schema metadata (pyarrow) or "
                f"config_meta (polars), or disable {POLARS_BIO_COORDINATE_SYSTEM_CHECK}.")
        glob = (get_option(POLARS_BIO_COORDINATE_SYSTEM_ZERO_BASED) or "false").lower() == "true"
        missing = [_type_name(d) for d, c in ((df1, cs1), (df2, cs2)) if c is None]
        warnings.warn(
            f"Coordinate system metadata is missing for: {', '.join(missing)}. "
            f"Using global POLARS_BIO_COORDINATE_SYSTEM_ZERO_BASED setting ({'0-based' if glob else '1-based'}).",
            UserWarning, stacklevel=4)
        cs1 = glob if cs1 is None else cs1
        cs2 = glob if cs2 is None else cs2
    if cs1 != cs2:
        s = lambda c: "0-based" if c else "1-based"
        raise CoordinateSystemMismatchError(
            f"Coordinate system mismatch: first input uses {s(cs1)} coordinates, "
            f"second input uses {s(cs2)} coordinates.")
    return cs1


def validate_coordinate_system_single(df) -> bool:
    """Single-input form (reference: polars_bio/_metadata.py validate_coordinate_system_single, used by
    merge / cluster / complement, range_op.py:87-111): returns zero_based."""
    cs = get_coordinate_system(df)
    if cs is None:
        check = (get_option(POLARS_BIO_COORDINATE_SYSTEM_CHECK) or "false").lower() == "true"
        if check:
            raise MissingCoordinateSystemError(
                f"{_type_name(df)} is missing coordinate system metadata.\n\n"
                f"Set df.attrs['{COORDINATE_SYSTEM_KEY}'] (pandas), schema metadata (pyarrow) or "
                f"config_meta (polars), or disable {POLARS_BIO_COORDINATE_SYSTEM_CHECK}.")
        cs = (get_option(POLARS_BIO_COORDINATE_SYSTEM_ZERO_BASED) or "false").lower() == "true"
        warnings.warn(
            f"Coordinate system metadata is missing for: {_type_name(df)}. "
            f"Using global POLARS_BIO_COORDINATE_SYSTEM_ZERO_BASED setting ({'0-based' if cs else '1-based'}).",
            UserWarning, stacklevel=4)
    return cs
