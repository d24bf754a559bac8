"""Configuration and comparison utilities (API of bioen/optimize/util.py)."""
import numpy as np
import yaml

from .ext import c_bioen


def library_gsl():
    return c_bioen.library_gsl()


def library_lbfgs():
    return c_bioen.library_lbfgs()


def compute_relative_difference_for_values(a, b):
    """|a-b|/|b|, or |a| when the reference value is zero (util.py:40-62)."""
    if b == 0:
        return abs(a)
    return abs(a - b) / abs(b)


def compute_relative_difference_for_arrays(a, b):
    """Largest element-wise relative difference over the non-zero entries of b, and its
    index among those entries (util.py:65-92)."""
    a = np.asarray(a)
    b = np.asarray(b)
    mask = b != 0.0
    if not mask.any():
        return 0.0, 0
    rel = np.abs(a[mask] - b[mask]) / np.abs(b[mask])
    idx = int(np.argmax(rel))
    return rel[idx], idx


def ntype(s):
    """String -> int, float, bool or unchanged string (util.py:163-188)."""
    for cast in (int, float):
        try:
            return cast(s)
        except Exception:
            pass
    low = s.lower()
    if low in ("true", "t", "yes", "y"):
        return True
    if low in ("false", "f", "no", "n"):
        return False
    return s


def nested_set(dic, keys, value):
    for key in keys[:-1]:
        dic = dic.setdefault(key, {})
    dic[keys[-1]] = value


def load_template_config_yaml(file_name, minimizer, parameter_mod=""):
    """yaml template -> the flat cfg dict the optimizers take (util.py:95-160):
    {minimizer, debug, verbose, params{...}, n_threads, cache_ytilde_transposed,
     algorithm, use_c_functions}.  `parameter_mod` is "a:b=v,c:d=v"."""
    minimizer = minimizer.lower()
    with open(file_name, "r") as fp:
        cfg = yaml.safe_load(fp)
    if parameter_mod:
        for token in parameter_mod.split(','):
            keys, value = token.split('=')
            nested_set(cfg, keys.split(':'), ntype(value))

    params = dict(cfg[minimizer])
    packed = {
        "minimizer": minimizer,
        "debug": cfg["general"]["debug"],
        "verbose": cfg["general"]["verbose"],
        "params": params,
        "n_threads": cfg["c_functions"]["n_threads"],
        "cache_ytilde_transposed": cfg["c_functions"]["cache_ytilde_transposed"],
        "algorithm": params.pop("algorithm", ""),
        "use_c_functions": params.pop("use_c_functions", True),
    }
    return packed
