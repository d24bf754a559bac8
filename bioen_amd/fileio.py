"""Pickle / HDF5 container IO with the interface of the reference's ``bioen/fileio.py``
(``load``, ``dump``, ``convert_to_hdf5`` and the lower-level ``load_pickle``, ``dump_pickle``,
``load_hdf5``, ``dump_hdf5``), plus the one step the device path adds: building a resident
``Context`` straight from an optimizer input file.

HDF5 goes through ``h5py`` when it is importable; this module never fakes it -- asking for an
``.h5`` file without h5py raises ImportError naming the missing package.  Pickle files (the format
of the reference's legacy test data, ``test/optimize/data/*.pkl``) need nothing.
"""
import os
import pickle
import string

import numpy as np

try:                                    # optional, exactly one place
    import h5py
except ImportError:                     # pragma: no cover - depends on the image
    h5py = None

# order of the entries in the reference's optimizer input files
# (test_find_opt_analytical_grad_logw.py:66-67, test_find_opt_analytical_grad_forces.py:60-61)
LOGW_KEYS = ["GInit", "G", "y", "yTilde", "YTilde", "w0", "theta"]
FORCES_KEYS = ["forces_init", "w0", "y", "yTilde", "YTilde", "theta"]


def _need_h5py(what):
    if h5py is None:
        raise ImportError("%s needs the h5py package, which is not installed (pickle files work without it)" % what)


def _kind(filename):
    ext = os.path.splitext(filename)[1]
    if ext not in (".pkl", ".h5"):
        raise ValueError("filename extension not recognized (only '.h5' or '.pkl')")    # fileio.py:41
    return ext


def load(filename, hdf5_deep_mode=False, hdf5_keys=[]):
    """Pickle: the stored object.  HDF5: the top-level datasets as a list (those named in
    ``hdf5_keys``, else all in sorted order), or the whole tree as nested dicts (``hdf5_deep_mode``)."""
    if _kind(filename) == ".pkl":
        return load_pickle(filename)
    return load_hdf5(filename, hdf5_deep_mode, hdf5_keys)


def dump(filename, data, hdf5_keys=[]):
    """Pickle: any object.  HDF5: a list/tuple (datasets named by ``hdf5_keys`` or 'AA', 'AB', ...)
    or a dict with string keys (nested dicts become groups)."""
    if _kind(filename) == ".pkl":
        dump_pickle(filename, data)
    else:
        dump_hdf5(filename, data, hdf5_keys)


def convert_to_hdf5(filename_pickle, filename_h5, hdf5_keys=[]):
    data = load_pickle(filename_pickle)
    if not isinstance(data, (list, tuple)):
        raise TypeError("the pickle file must hold a flat list of arrays / scalars")
    dump_hdf5(filename_h5, data, hdf5_keys)


def load_pickle(file_name):
    with open(file_name, "rb") as fp:
        return pickle.load(fp)


def dump_pickle(file_name, data):
    with open(file_name, "wb") as fp:
        pickle.dump(data, fp)


def _read_group(group):
    out = {}
    for key in sorted(group.keys()):
        item = group[key]
        out[key] = item[()] if isinstance(item, h5py.Dataset) else _read_group(item)
    return out


def load_hdf5(file_name, hdf5_deep_mode=False, hdf5_keys=[]):
    _need_h5py("reading " + file_name)
    with h5py.File(file_name, "r") as f:
        if hdf5_deep_mode:
            return _read_group(f)
        names = list(hdf5_keys) if hdf5_keys else sorted(f.keys())
        return [f[k][()] for k in names if isinstance(f[k], h5py.Dataset)]


def _label(i):
    """'AA', 'AB', ..., 'ZZ': sortable stand-in names for unlabeled list entries (fileio.py:148-156)"""
    n = len(string.ascii_uppercase)
    return string.ascii_uppercase[i // n] + string.ascii_uppercase[i % n]


def _write_group(group, data):
    for key, value in data.items():
        if isinstance(value, dict):
            _write_group(group.create_group(key), value)
        else:
            group.create_dataset(key, data=value)


def dump_hdf5(file_name, data, data_labels=[]):
    _need_h5py("writing " + file_name)
    with h5py.File(file_name, "w") as f:
        if isinstance(data, (list, tuple)):
            labels = list(data_labels) if len(data_labels) == len(data) else [_label(i) for i in range(len(data))]
            for name, value in zip(labels, data):
                f.create_dataset(name, data=value)
        elif isinstance(data, dict):
            _write_group(f, data)
        else:
            raise TypeError("data type unsupported")


# ---- device side ---------------------------------------------------------------------------------
def load_optimizer_input(filename, method="log_weights"):
    """The reference's optimizer input file as a dict keyed by LOGW_KEYS / FORCES_KEYS."""
    keys = LOGW_KEYS if method == "log_weights" else FORCES_KEYS
    values = load(filename, hdf5_keys=keys) if _kind(filename) == ".h5" else load(filename)
    if len(values) != len(keys):
        raise ValueError("%s: expected %d entries (%s), found %d" % (filename, len(keys), ", ".join(keys), len(values)))
    return dict(zip(keys, values))


def context_from_file(filename, method="log_weights", device=0):
    """-> (Context with yTilde resident in HBM, dict of the remaining inputs).  The matrix is handed
    to the device in the file's own memory layout when that is C-contiguous float64 (no host copy)."""
    from . import Context
    d = load_optimizer_input(filename, method)
    yTilde = np.asarray(d["yTilde"], dtype=np.float64)
    ctx = Context(yTilde, np.asarray(d["YTilde"], dtype=np.float64).ravel(), device=device)
    return ctx, d
