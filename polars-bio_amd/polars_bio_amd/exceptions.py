"""Exceptions mirrored from the reference (/root/reference/polars_bio/exceptions.py:4,23)."""


class CoordinateSystemMismatchError(Exception):
    """The two inputs of a range operation use different coordinate systems
    (one 0-based half-open, the other 1-based closed)."""


class MissingCoordinateSystemError(Exception):
    """An input lacks coordinate-system metadata and
    ``datafusion.bio.coordinate_system_check`` is "true"."""
