"""String key/value option store (reference: polars_bio/context.py:29-69,
src/context.rs:35-54).  Only the keys that reach the range-operation hot path
have an effect; the rest are stored and returned verbatim."""
from __future__ import annotations

import numbers
import threading

from .constants import POLARS_BIO_COORDINATE_SYSTEM_CHECK, POLARS_BIO_COORDINATE_SYSTEM_ZERO_BASED


class Context:
    def __init__(self):
        self._lock = threading.Lock()
        self._opts = {
            # reference defaults: polars_bio/context.py:33-50
            "datafusion.execution.target_partitions": "1",
            "datafusion.execution.batch_size": "8192",
            POLARS_BIO_COORDINATE_SYSTEM_ZERO_BASED: "false",
            POLARS_BIO_COORDINATE_SYSTEM_CHECK: "false",
            "bio.interval_join_algorithm": "hip",
            # engine options of this implementation
            "ivj.device": "auto",   # "auto": LOCAL_RANK (one process per GPU) or 0; a number pins the device
            # several GPUs in ONE process (multi.MultiEngine: one context + host thread per device, contigs dealt out):
            # "ivj.devices" = explicit slots ("0,1"); else ivj.num_gpus devices counted from ivj.device, capped by the visible
            # devices.  datafusion.execution.target_partitions (the reference's parallelism knob, polars_bio/context.py:36) is
            # stored for call compatibility only: raising it must not change which GPUs a process touches
            "ivj.devices": "auto",
            "ivj.num_gpus": "0",
            "ivj.low_memory_batch_rows": "8000000",
            # joined rows of pb.overlap: the key columns of both sides always come back from HBM in pair order
            # (ivj_overlap_rows); the other columns are gathered by the pair indices on the host ("host", native threaded
            # gather) or through HBM ("device", ivj_take); "pairs": index pairs only, every column gathered on the host
            "ivj.materialize": "host",
        }

    def set_option(self, key, value):
        if isinstance(value, bool):
            value = "true" if value else "false"
        elif isinstance(value, numbers.Number):
            value = str(value)
        with self._lock:
            changed = self._opts.get(key) != value
            self._opts[key] = value
        if changed and key in ("ivj.device", "ivj.devices", "ivj.num_gpus"):
            from ._engine import reset_default_engine      # the next call builds the engine(s) the new value asks for
            reset_default_engine()

    def get_option(self, key):
        with self._lock:
            return self._opts.get(key)

    def sync_options(self):  # kept for call-shape parity (range_op_helpers.py:181)
        return None


ctx = Context()


def set_option(key, value):
    ctx.set_option(key, value)


def get_option(key):
    return ctx.get_option(key)
