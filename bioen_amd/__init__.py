"""bioen_amd -- MI355X-native implementation of BioEn's optimizer hot path.

    from bioen_amd import optimize            # drop-in for ``from bioen import optimize``
    from bioen_amd import Context             # device-resident problem (C ABI wrapper)
    from bioen_amd import sweep               # theta-series sharded over GPUs
"""
from ._lib import Context, BioenHipError, device_count, LIB_PATH  # noqa: F401
from . import optimize  # noqa: F401
from . import sweep  # noqa: F401
from . import nuisance  # noqa: F401

__version__ = "0.1.0"
