"""Minimizer configuration front end (API of bioen/optimize/minimize.py)."""
from __future__ import print_function

import os

from . import util
from .ext import c_bioen


def set_fast_openmp_flag(flag):
    c_bioen.set_fast_openmp_flag(flag)


def get_fast_openmp_flag():
    return c_bioen.get_fast_openmp_flag()


def show_params(packed_params):
    for key in ("minimizer", "verbose", "params", "algorithm", "use_c_functions", "n_threads",
                "cache_ytilde_transposed"):
        print("%-24s" % key, packed_params[key])
    print("------------------------------")


def Parameters(minimizer, parameter_mod=""):
    """Default parameter dict of `minimizer` in {"lbfgs", "gsl", "scipy"} (minimize.py:44-66)."""
    template = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config", "bioen_optimize.yaml")
    if not os.path.isfile(template):
        print("Default parameter file (", template, ") cannot be found!")
    return util.load_template_config_yaml(template, minimizer, parameter_mod)
