"""Defaults mirrored from /root/reference/polars_bio/constants.py:1-10."""
DEFAULT_INTERVAL_COLUMNS = ["chrom", "start", "end"]
DEFAULT_BATCH_SIZE = 8192
POLARS_BIO_COORDINATE_SYSTEM_ZERO_BASED = "datafusion.bio.coordinate_system_zero_based"
POLARS_BIO_COORDINATE_SYSTEM_CHECK = "datafusion.bio.coordinate_system_check"
COORDINATE_SYSTEM_KEY = "coordinate_system_zero_based"
