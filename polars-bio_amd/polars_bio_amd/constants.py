"""Names the front end shares with the reference's option keys and defaults.

The VALUES are part of the drop-in contract (a user's ``pb.set_option("datafusion.bio...", ...)`` calls and the
metadata key written by the reference's readers must keep working), so they equal
/root/reference/polars_bio/constants.py:1-10; nothing else is taken from there.
"""

# metadata key under which a frame carries its coordinate system (pandas attrs, polars config_meta, Arrow schema)
COORDINATE_SYSTEM_KEY = "coordinate_system_zero_based"

# session options consulted by validate_coordinate_systems (see _metadata.py)
_OPTION_PREFIX = "datafusion.bio."
POLARS_BIO_COORDINATE_SYSTEM_CHECK = _OPTION_PREFIX + "coordinate_system_check"            # "true": missing metadata raises
POLARS_BIO_COORDINATE_SYSTEM_ZERO_BASED = _OPTION_PREFIX + "coordinate_system_zero_based"  # fallback when it is missing

# column names assumed when cols1 / cols2 are None, and the reference's streaming batch size
DEFAULT_INTERVAL_COLUMNS = ["chrom", "start", "end"]
DEFAULT_BATCH_SIZE = 8192
