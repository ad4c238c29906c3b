"""Device-resident range operations on torch CUDA(=HIP) tensors.

torch is plumbing here (device memory, the current stream, torch.distributed); the
work is done by libivjoin_hip.so through the ``*_dev`` entry points of include/ivjoin.h.
Columns are int32 torch tensors already resident in HBM; results stay in HBM.

Import order: torch wheels bundle their own ROCm runtime; in a process that uses both, import
torch BEFORE the first ``Engine`` is created (this module does so itself), otherwise torch may not
see its GPUs ("No HIP GPUs are available").
"""

from __future__ import annotations

import torch  # noqa: F401  (must precede the dlopen of libivjoin_hip.so in this process)

from typing import Optional, Tuple

from ._engine import Engine, make_opts


class DeviceSide:
    """(contig, start, end[, row_id]) int32 CUDA tensors of one side."""

    def __init__(self, contig, start, end, row_id=None):
        import torch
        for t in (contig, start, end) + ((row_id,) if row_id is not None else ()):
            if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous():
                raise ValueError("columns must be contiguous int32 CUDA tensors")
        self.contig, self.start, self.end, self.row_id = contig, start, end, row_id
        self.n = int(contig.shape[0])

    def as_c(self):
        return Engine.dev_side(self.contig.data_ptr(), self.start.data_ptr(), self.end.data_ptr(), self.n,
                               self.row_id.data_ptr() if self.row_id is not None else 0)


class DeviceJoin:
    """One engine bound to torch's current stream on ``device``."""

    def __init__(self, device: int = 0):
        import torch
        self.torch = torch
        self.device = device
        torch.cuda.set_device(device)
        self.engine = Engine(device)
        self.engine.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def build_index(self, build: DeviceSide, strict: bool, n_contigs: int, with_end_order: bool = False):
        return self.engine.index_build_dev(build.as_c(), make_opts(strict, n_contigs), with_end_order)

    def overlap(self, probe: DeviceSide, build: DeviceSide, strict: bool, n_contigs: int, index=None, out=None,
                fused: bool = True, partition_mode: int = 0):
        """Index build (radix sort) + count + scan + fill.  -> (probe_idx, build_idx) int32 tensors.
        ``out``: optional pair of preallocated int32 CUDA tensors; views of their first n_pairs
        elements are returned when they are large enough (no allocation on the call path).  With
        ``out`` and ``fused`` the single-pass ivj_overlap_fused_dev is tried first."""
        torch = self.torch
        opts = make_opts(strict, n_contigs, partition_mode=partition_mode)
        own = index is None
        ix = self.engine.index_build_dev(build.as_c(), opts, False) if own else index
        try:
            side = probe.as_c()
            if out is not None and fused:
                # single fused pass into the caller's buffers (the library picks the window-scan kernel for
                # sparse results and the flat candidate kernel when the buffers say >= 16 pairs per probe)
                cap = min(out[0].numel(), out[1].numel())
                total, fits = self.engine.overlap_fused_dev(ix, side, opts, out[0].data_ptr(), out[1].data_ptr(), cap)
                if fits:
                    return out[0][:total], out[1][:total]
            total = self.engine.overlap_count_dev(ix, side, opts)
            if out is not None and out[0].numel() >= total and out[1].numel() >= total:
                out_p, out_b = out[0][:total], out[1][:total]
            else:
                out_p = torch.empty(total, dtype=torch.int32, device=probe.start.device)
                out_b = torch.empty(total, dtype=torch.int32, device=probe.start.device)
            self.engine.overlap_fill_dev(ix, side, opts, out_p.data_ptr(), out_b.data_ptr(), total)
        finally:
            if own:
                ix.close()
        return out_p, out_b

    def overlap_rows(self, probe: DeviceSide, build: DeviceSide, strict: bool, n_contigs: int, out: dict, index=None,
                     partition_mode: int = 0):
        """Join + row materialisation in one pass (ivj_overlap_fused_rows_dev) into the preallocated int32
        CUDA tensors of ``out`` (keys from _engine.ROW_COLUMNS; a missing key = column not wanted).
        -> (dict of views of the first n_rows elements, n_rows, fits); fits=False: grow ``out`` to n_rows."""
        opts = make_opts(strict, n_contigs, partition_mode=partition_mode)
        own = index is None
        ix = self.engine.index_build_dev(build.as_c(), opts, False) if own else index
        try:
            cap = min(int(t.numel()) for t in out.values())
            ptrs = {f"{k}_ptr": t.data_ptr() for k, t in out.items()}
            total, fits = self.engine.overlap_fused_rows_dev(ix, probe.as_c(), opts, cap, **ptrs)
        finally:
            if own:
                ix.close()
        return ({k: t[:total] for k, t in out.items()} if fits else {}), total, fits

    def materialize(self, probe: DeviceSide, build: DeviceSide, probe_idx, build_idx, out=None):
        """Row materialisation (ivj_materialize_dev): for every pair the key columns of both sides.
        -> dict contig / start_1 / end_1 / start_2 / end_2 of int32 CUDA tensors (``out``: optional dict
        of preallocated tensors of at least n_pairs elements)."""
        torch = self.torch
        n = int(probe_idx.shape[0])
        names = ("contig", "start_1", "end_1", "start_2", "end_2")
        cols = {k: (out[k][:n] if out is not None else torch.empty(n, dtype=torch.int32, device=probe.start.device)) for k in names}
        self.engine.materialize_dev(probe.as_c(), build.as_c(), n, probe_idx.data_ptr(), build_idx.data_ptr(),
                                    *(cols[k].data_ptr() for k in names))
        return cols

    def take(self, column, idx, with_validity: bool = False):
        """Arrow take of one 4- or 8-byte CUDA column by int32 row indices (negative -> 0 / null).
        -> values, or (values, validity bitmap as int64 words) with ``with_validity``."""
        torch = self.torch
        if column.element_size() not in (4, 8) or not column.is_contiguous():
            raise ValueError("column must be a contiguous tensor of 4- or 8-byte elements")
        n = int(idx.shape[0])
        dst = torch.empty(n, dtype=column.dtype, device=column.device)
        val = torch.zeros((n + 63) // 64, dtype=torch.int64, device=column.device) if with_validity else None
        self.engine.take_dev(column.data_ptr(), column.element_size(), idx.data_ptr(), n, dst.data_ptr(),
                             val.data_ptr() if val is not None else 0)
        return (dst, val) if with_validity else dst

    def count_overlaps(self, probe: DeviceSide, build: DeviceSide, strict: bool, n_contigs: int, index=None,
                       partition_mode: int = 0):
        torch = self.torch
        opts = make_opts(strict, n_contigs, partition_mode=partition_mode)
        own = index is None
        ix = self.engine.index_build_dev(build.as_c(), opts, True) if own else index
        try:
            out = torch.empty(probe.n, dtype=torch.int64, device=probe.start.device)
            self.engine.count_overlaps_dev(ix, probe.as_c(), opts, out.data_ptr())
        finally:
            if own:
                ix.close()
        return out

    # ---- sort-scan family ---------------------------------------------
    def coverage(self, probe: DeviceSide, build: DeviceSide, strict: bool, n_contigs: int, index=None, out=None):
        """Covered positions of every probe row by the union of the build side -> int64 tensor."""
        torch = self.torch
        opts = make_opts(strict, n_contigs)
        own = index is None
        ix = self.engine.index_build_dev(build.as_c(), opts, False) if own else index
        try:
            cov = out if out is not None else torch.empty(probe.n, dtype=torch.int64, device=probe.start.device)
            self.engine.coverage_dev(ix, probe.as_c(), opts, cov.data_ptr())
        finally:
            if own:
                ix.close()
        return cov

    def merge(self, frame: DeviceSide, strict: bool, n_contigs: int, min_dist: int = 0, out=None):
        """Merged intervals of one frame -> (contig, start, end int32, n_intervals int64) tensors.
        ``out``: optional preallocated 4-tuple (views of the first n_merged elements are returned)."""
        torch = self.torch
        opts = make_opts(strict, n_contigs)
        ix = self.engine.index_build_dev(frame.as_c(), opts, False, sweep_only=True)
        try:
            if out is None:
                dev = frame.start.device
                out = tuple(torch.empty(frame.n, dtype=dt, device=dev) for dt in (torch.int32, torch.int32, torch.int32, torch.int64))
            n, fits = self.engine.merge_dev(ix, opts, min_dist, min(int(t.numel()) for t in out), *(t.data_ptr() for t in out))
            if not fits:
                raise ValueError(f"merge output buffers hold fewer than {n} intervals")
        finally:
            ix.close()
        return tuple(t[:n] for t in out)

    def subtract(self, left: DeviceSide, right: DeviceSide, strict: bool, n_contigs: int, index=None, out=None):
        """left minus the union of right -> (left row, start, end) int32 tensors of the remaining pieces."""
        torch = self.torch
        opts = make_opts(strict, n_contigs)
        own = index is None
        ix = self.engine.index_build_dev(right.as_c(), opts, False) if own else index
        try:
            if out is not None:
                n, fits = self.engine.subtract_dev(ix, left.as_c(), opts, min(int(t.numel()) for t in out), *(t.data_ptr() for t in out))
                if fits:
                    return tuple(t[:n] for t in out)
            else:
                n, _ = self.engine.subtract_dev(ix, left.as_c(), opts, 0, 0, 0, 0)
            out = tuple(torch.empty(n, dtype=torch.int32, device=left.start.device) for _ in range(3))
            n, fits = self.engine.subtract_dev(ix, left.as_c(), opts, n, *(t.data_ptr() for t in out))
        finally:
            if own:
                ix.close()
        return out

    def nearest(self, probe: DeviceSide, build: DeviceSide, strict: bool, n_contigs: int, k: int = 1,
                include_overlaps: bool = True, index=None, partition_mode: int = 0):
        torch = self.torch
        opts = make_opts(strict, n_contigs, k, include_overlaps, partition_mode=partition_mode)
        own = index is None
        general = not (k == 1 and include_overlaps)
        ix = self.engine.index_build_dev(build.as_c(), opts, general) if own else index
        try:
            dev = probe.start.device
            idx = torch.empty((probe.n, k), dtype=torch.int32, device=dev)
            dist = torch.empty((probe.n, k), dtype=torch.int64, device=dev)
            nf = torch.empty(probe.n, dtype=torch.int32, device=dev)
            self.engine.nearest_dev(ix, probe.as_c(), opts, idx.data_ptr(), dist.data_ptr(), nf.data_ptr())
        finally:
            if own:
                ix.close()
        return idx, dist, nf
