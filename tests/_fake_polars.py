"""A stand-in for the few polars names the lazy glue touches (polars_bio_amd/_polars_lazy.py), for an image without polars:
DataFrame / LazyFrame over pyarrow tables, ``from_arrow``, ``io.plugins.register_io_source`` with the IO-plugin protocol
(``source(with_columns, predicate, n_rows, batch_size) -> iterator of DataFrames``, projection / predicate / row limit pushed
into the source the way polars' optimiser does) and ``LazyFrame.collect_batches(lazy=True, engine="streaming")`` whose
``_inner`` is an Arrow C stream.  Test infrastructure only; the tests of the real thing are gated on ``importorskip("polars")``."""
import sys
import types

import pyarrow as pa
import pyarrow.compute as pc


class Schema(dict):
    def __init__(self, arrow_schema):
        super().__init__({f.name: f.type for f in arrow_schema})
        self._arrow = arrow_schema

    def to_arrow(self):
        return self._arrow


class DataFrame:
    def __init__(self, data=None, schema=None):
        if isinstance(data, pa.Table):
            self._t = data
        elif data is None and schema is not None:
            self._t = (schema._arrow if isinstance(schema, Schema) else pa.schema(list(schema.items()))).empty_table()
        else:
            self._t = pa.table(data)

    def to_arrow(self):
        return self._t

    @property
    def height(self):
        return self._t.num_rows

    @property
    def columns(self):
        return self._t.column_names

    @property
    def schema(self):
        return Schema(self._t.schema)

    def head(self, n=5):
        return DataFrame(self._t.slice(0, n))

    def filter(self, predicate):                    # a "predicate expression" here = callable(pa.Table) -> boolean mask
        return DataFrame(self._t.filter(predicate(self._t)))

    def select(self, cols):
        return DataFrame(self._t.select(list(cols)))

    def lazy(self):
        t = self._t
        return LazyFrame(lambda wc, pred, n, bs: iter([DataFrame(t)]), Schema(t.schema))

    def __len__(self):
        return self._t.num_rows


class _Batches:
    def __init__(self, it, schema):
        self._it = it
        self._inner = pa.RecordBatchReader.from_batches(schema, (rb for df in it for rb in df.to_arrow().cast(schema).to_batches()))

    def __iter__(self):
        return self._it


class LazyFrame:
    """source(with_columns, predicate, n_rows, batch_size) -> iterator of DataFrames (the IO-plugin protocol)."""

    def __init__(self, source, schema, with_columns=None, predicate=None, n_rows=None):
        self._source, self._schema = source, schema
        self._wc, self._pred, self._n = with_columns, predicate, n_rows
        self.runs = 0                                 # how many times the source was started (fresh stream per collect)

    def collect_schema(self):
        if self._wc is not None:
            return Schema(pa.schema([self._schema._arrow.field(c) for c in self._wc]))
        return self._schema

    def _run(self, batch_size=None):
        self.runs += 1
        left = self._n
        for df in self._source(self._wc, self._pred, self._n, batch_size):
            # polars re-applies what it pushed down: a source may ignore the hints
            if self._pred is not None and not getattr(df, "_filtered", False):
                pass
            if left is not None:
                if df.height > left:
                    df = df.head(left)
                left -= df.height
            yield df
            if left is not None and left <= 0:
                return

    def collect(self):
        parts = [df.to_arrow() for df in self._run()]
        sch = self.collect_schema()._arrow
        return DataFrame(pa.concat_tables([p.cast(sch) for p in parts]) if parts else sch.empty_table())

    def collect_batches(self, lazy=True, engine="streaming", chunk_size=None):
        return _Batches(self._run(chunk_size), self.collect_schema()._arrow)

    def head(self, n=5):
        return LazyFrame(self._source, self._schema, self._wc, self._pred, n if self._n is None else min(n, self._n))

    limit = head

    def select(self, cols):
        return LazyFrame(self._source, self._schema, list(cols), self._pred, self._n)

    def filter(self, predicate):
        return LazyFrame(self._source, self._schema, self._wc, predicate, self._n)

    def explain(self):
        return "PYTHON SCAN []"


def from_arrow(t):
    return DataFrame(t if isinstance(t, pa.Table) else pa.Table.from_batches([t]))


def register_io_source(io_source, schema=None, **_):
    return LazyFrame(io_source, schema if isinstance(schema, Schema) else Schema(pa.schema(list(schema.items()))))


def install(monkeypatch):
    """Make ``import polars`` / ``from polars.io.plugins import register_io_source`` resolve to this module and tell the
    package's modules that polars is there."""
    me = sys.modules[__name__]
    io = types.ModuleType("polars.io")
    plugins = types.ModuleType("polars.io.plugins")
    plugins.register_io_source = register_io_source
    io.plugins = plugins
    monkeypatch.setitem(sys.modules, "polars", me)
    monkeypatch.setitem(sys.modules, "polars.io", io)
    monkeypatch.setitem(sys.modules, "polars.io.plugins", plugins)
    monkeypatch.setattr(me, "io", io, raising=False)
    from polars_bio_amd import _arrow, _metadata
    monkeypatch.setattr(_arrow, "pl", me)
    monkeypatch.setattr(_metadata, "pl", me)
    return me
