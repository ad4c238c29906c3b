"""Helpers shared by the test modules (fixture loading, synthetic intervals)."""
import csv
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def read_csv_cols(path):
    with open(path, newline="") as f:
        rows = list(csv.reader(f))
    hdr, body = rows[0], [r for r in rows[1:] if r]
    return {h: [r[i] for r in body] for i, h in enumerate(hdr)}


def load_intervals_csv(path):
    t = read_csv_cols(path)
    return t["contig"], np.array(t["pos_start"], np.int64), np.array(t["pos_end"], np.int64)


def load_cases():
    with open(os.path.join(GOLDEN, "cases.json")) as f:
        return json.load(f)


def load_parquet_intervals(name):
    import pyarrow.parquet as pq
    t = pq.read_table(os.path.join(GOLDEN, name))
    return (t.column("contig").to_pylist(),
            t.column("pos_start").to_numpy().astype(np.int64),
            t.column("pos_end").to_numpy().astype(np.int64))


def random_side(rng, n, n_contigs, span, max_len, zero_len_frac=0.05, dup_frac=0.1):
    """Random intervals with duplicates, zero-length rows and heavy nesting."""
    c = rng.integers(0, n_contigs, n).astype(np.int32)
    s = rng.integers(0, span, n).astype(np.int32)
    ln = rng.integers(0, max_len + 1, n).astype(np.int32)
    ln[rng.random(n) < zero_len_frac] = 0
    long_mask = rng.random(n) < 0.02
    ln[long_mask] = rng.integers(0, span, long_mask.sum())
    e = (s + ln).astype(np.int32)
    if n > 1:
        d = rng.random(n) < dup_frac
        src = rng.integers(0, n, n)
        c[d], s[d], e[d] = c[src[d]], s[src[d]], e[src[d]]
    return c, s, e


class OracleEngine:
    """Test double with the Engine host API, backed by the CPU oracle.  Lets the
    CPU suite exercise the front end's host logic (key encoding, result assembly,
    metadata) without a GPU.  Never used by the product path."""

    def overlap(self, probe, build, strict, n_contigs):
        from oracle import oracle as O
        b = O.Side(*build)
        return O.overlap_fast(O.Index(b, n_contigs), O.Side(*probe), strict)

    def take_columns(self, idx, columns, nullable=False):
        import numpy as np
        idx = np.asarray(idx, np.int32)
        out = []
        for c in columns:
            c = np.asarray(c)
            v = np.where(idx >= 0, c[np.where(idx >= 0, idx, 0)], 0).astype(c.dtype) if len(c) else np.zeros(len(idx), c.dtype)
            val = None
            if nullable:
                bits = np.zeros(((len(idx) + 63) // 64) * 64, np.uint8)
                bits[:len(idx)] = idx >= 0
                val = np.packbits(bits, bitorder="little").view(np.uint64)
            out.append((v, val))
        return out

    def overlap_rows(self, probe, build, strict, n_contigs, partition_mode=0, as_arrow=False):
        import numpy as np
        import pyarrow as pa
        p, b = self.overlap(probe, build, strict, n_contigs)
        cols = {"probe_idx": p, "build_idx": b, "contig": probe[0][p], "start_1": probe[1][p], "end_1": probe[2][p],
                "start_2": build[1][b], "end_2": build[2][b]}
        cols = {k: np.ascontiguousarray(v, np.int32) for k, v in cols.items()}
        return pa.record_batch(cols) if as_arrow else cols

    def merge(self, frame, strict, n_contigs, min_dist=0):
        import numpy as np
        from oracle import oracle as O
        _, _, _, (c, s, e, n) = O.np_cluster(O.Side(*frame), strict, min_dist)
        return c, s.astype(np.int32), e.astype(np.int32), n

    def cluster(self, frame, strict, n_contigs, min_dist=0):
        import numpy as np
        from oracle import oracle as O
        cid, cs, ce, merged = O.np_cluster(O.Side(*frame), strict, min_dist)
        return cid, cs.astype(np.int32), ce.astype(np.int32), len(merged[0])

    def coverage(self, probe, build, strict, n_contigs):
        from oracle import oracle as O
        return O.np_coverage_fast(O.Side(*probe), O.Side(*build), strict)

    def subtract(self, left, right, strict, n_contigs):
        import numpy as np
        from oracle import oracle as O
        r, s, e = O.np_subtract(O.Side(*left), O.Side(*right), strict)
        return r, s.astype(np.int32), e.astype(np.int32)

    def complement(self, frame, view, strict, n_contigs):
        return self.subtract(view, frame, strict, n_contigs)

    def overlap_batches(self, probe, build, strict, n_contigs, batch_rows=8_000_000):
        import numpy as np
        from oracle import oracle as O
        ix = O.Index(O.Side(*build), n_contigs)
        n = len(probe[0])
        for lo in range(0, n, batch_rows):
            hi = min(lo + batch_rows, n)
            p, b = O.overlap_fast(ix, O.Side(probe[0][lo:hi], probe[1][lo:hi], probe[2][lo:hi]), strict)
            yield (p + lo).astype(np.int32), b

    def probe_stream(self, build, strict, n_contigs, op=0, max_batch_rows=8_000_000, k=1, include_overlaps=True, partition_mode=0, copy=True):
        return _OracleStream(build, strict, n_contigs, op, k, include_overlaps)

    def count_overlaps(self, probe, build, strict, n_contigs):
        from oracle import oracle as O
        b = O.Side(*build)
        return O.count_overlaps_fast(O.Index(b, n_contigs), O.Side(*probe), strict)

    def nearest(self, probe, build, strict, n_contigs, k=1, include_overlaps=True):
        from oracle import oracle as O
        b = O.Side(*build)
        return O.nearest_fast(O.Index(b, n_contigs), O.Side(*probe), strict, k, include_overlaps)


class _OracleStream:
    """Test double of _engine.ProbeStream: same protocol (results of a batch arrive two calls after its submit), the
    per-batch answers come from the CPU oracle."""

    def __init__(self, build, strict, n_contigs, op, k, include_overlaps):
        from oracle import oracle as O
        self.O, self.ix = O, O.Index(O.Side(*build), n_contigs)
        self.strict, self.op, self.k, self.inc = strict, op, k, include_overlaps
        self.queue, self.n = [], 0

    def _answer(self, batch):
        O, side = self.O, self.O.Side(*batch)
        out = {"batch": self.n, "n_probe": side.n}
        if self.op == 0:
            out["probe_idx"], out["build_idx"] = O.overlap_fast(self.ix, side, self.strict)
        elif self.op == 1:
            out["counts"] = O.count_overlaps_fast(self.ix, side, self.strict)
        else:
            out["build_idx"], out["dist"], out["n_found"] = O.nearest_fast(self.ix, side, self.strict, self.k, self.inc)
        self.n += 1
        return out

    def _hand_out(self, res):
        """Like the real session with copy=False: the arrays handed out are views of ONE recycled slot, and the previous hand-out is
        scribbled over -- a front end that keeps a result array (instead of gathering through it or copying it) shows up here."""
        import numpy as np
        for a in getattr(self, "_lent", ()):
            a[...] = -7
        lent, out = [], {}
        for key, v in res.items():
            if isinstance(v, np.ndarray) and not (self.op == 1 and key == "counts"):    # (counts are handed out as copies)
                v = v.copy()
                lent.append(v)
            out[key] = v
        self._lent = lent
        return out

    def submit(self, batch):
        self.queue.append(self._answer(batch))
        return self._hand_out(self.queue.pop(0)) if len(self.queue) > 2 else None

    def flush(self):
        return self._hand_out(self.queue.pop(0)) if self.queue else None

    def close(self):
        self.queue = []
