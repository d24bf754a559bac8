"""``bioen_amd.optimize`` -- the ``bioen.optimize`` API (bioen/optimize/__init__.py:1-11)
on MI355X: same modules, same call signatures, numerics in hand-written HIP kernels."""
from . import common
from . import util
from . import minimize
from . import forces
from . import log_weights
from .ext.c_bioen import hold as resident      # ``with optimize.resident(yTilde): <theta loop>``: one upload for the block

__all__ = ["common", "util", "minimize", "forces", "log_weights", "resident"]
