"""bioen_amd -- MI355X-native implementation of BioEn's optimizer hot path.

    from bioen_amd import optimize            # drop-in for ``from bioen import optimize``
    from bioen_amd import Context             # device-resident problem (C ABI wrapper)
    from bioen_amd import sweep               # theta-series sharded over GPUs
"""
from ._lib import Context, BioenHipError, device_count, LIB_PATH  # noqa: F401
from . import optimize  # noqa: F401
from . import sweep  # noqa: F401
from . import nuisance  # noqa: F401

__version__ = "0.4.0"


def install_as_bioen():
    """Make ``from bioen import optimize`` -- the import of the reference's callers (``bioen/analyze/procedure.py:9``:
    ``from .. import optimize``; its tests and notebooks: ``from bioen import optimize``) -- resolve to this package, so
    that they run unchanged on the MI355X path.  Call it once, before those callers are imported.

    If a BioEn source tree or installation is importable, its ``bioen`` package is kept for everything else
    (``bioen.analyze``, ``bioen.fileio`` ...) -- located WITHOUT being executed, since its ``__init__`` would import the
    Cython extension this package replaces -- and only ``bioen.optimize`` (with its submodules ``log_weights``,
    ``forces``, ``minimize``, ``common``, ``util``, ``ext.c_bioen``) is taken over.  Otherwise a bare ``bioen`` package
    holding nothing but ``optimize`` is registered.  Returns the ``bioen`` module."""
    import importlib.util
    import sys
    import types

    pkg = sys.modules.get("bioen")
    if pkg is None:
        pkg = types.ModuleType("bioen")
        pkg.__doc__ = "bioen: optimize = bioen_amd.optimize (bioen_amd.install_as_bioen)"
        try:
            spec = importlib.util.find_spec("bioen")
        except (ImportError, ValueError):
            spec = None
        pkg.__path__ = list(spec.submodule_search_locations) if spec and spec.submodule_search_locations else []
        pkg.__package__ = "bioen"
        sys.modules["bioen"] = pkg
    pkg.optimize = optimize
    sys.modules["bioen.optimize"] = optimize
    for name in ("log_weights", "forces", "minimize", "common", "util", "ext"):
        sys.modules["bioen.optimize." + name] = getattr(optimize, name)
    sys.modules["bioen.optimize.ext.c_bioen"] = optimize.ext.c_bioen
    return pkg
