"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol
include/bioen_hip.h declares, the ctypes struct layouts match the header, and -- with no
GPU in this container -- compute entry points fail LOUDLY instead of falling back."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, LBFGS_DEFAULTS

HEADER = os.path.join(ROOT, "include", "bioen_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bioen_hip_\w+)\s*\(", src)))


def test_library_is_built_and_loads():
    from bioen_amd import _lib
    assert os.path.isfile(_lib.LIB_PATH), "run `make -C bioen_amd/csrc` (graft build())"
    L = _lib.lib()
    assert b"gfx950" in L.bioen_hip_version()


def test_every_declared_symbol_is_exported_and_bound():
    from bioen_amd import _lib
    names = declared_functions()
    assert len(names) >= 25
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(line.split()[-1] for line in out.splitlines() if line.strip())
    missing = [n for n in names if n not in exported]
    assert not missing, "declared in bioen_hip.h but not exported: %s" % missing
    unbound = [n for n in names if n not in _lib.exported_symbols()]
    assert not unbound, "declared in bioen_hip.h but not bound in _lib.py: %s" % unbound
    extra = [n for n in _lib.exported_symbols() if n not in names]
    assert not extra, "bound in _lib.py but not declared in the header: %s" % extra


def test_struct_layouts_match_the_reference_packing():
    from bioen_amd import _lib
    # lbfgs_config_params of c_bioen_common.h:69-79: int,int,5 doubles,int,int -> 56 bytes on LP64
    assert C.sizeof(_lib.LbfgsConfig) == 56
    assert _lib.LbfgsConfig.delta.offset == 8 and _lib.LbfgsConfig.past.offset == 48
    assert C.sizeof(_lib.VisualParams) == 16
    assert C.sizeof(_lib.OptResult) == 48
    from oracle import ref_binding
    assert C.sizeof(ref_binding.lbfgs_config_params) == C.sizeof(_lib.LbfgsConfig)


def test_error_strings_match_reference_texts():
    from bioen_amd import _lib
    e = np.load(os.path.join(ROOT, "tests", "golden", "error_codes.npz"))
    L = _lib.lib()
    assert L.bioen_hip_lbfgs_strerror(-1015).decode() == str(e["msg_delta_neg"])
    assert L.bioen_hip_lbfgs_strerror(-997).decode() == str(e["msg_maxiter"])
    assert L.bioen_hip_lbfgs_strerror(0).decode() == str(e["msg_0"])
    assert L.bioen_hip_lbfgs_strerror(1).decode() == str(e["msg_1"])
    assert L.bioen_hip_lbfgs_strerror(2).decode() == str(e["msg_2"])


def _no_gpu():
    from bioen_amd import _lib
    return _lib.device_count() == 0


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is present; the no-device behaviour cannot be observed")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    import bioen_amd
    from bioen_amd.optimize import log_weights, forces, minimize
    yT = np.random.default_rng(0).normal(size=(4, 6))
    YT = np.zeros((1, 4))
    with pytest.raises(bioen_amd.BioenHipError) as exc:
        bioen_amd.Context(yT, YT)
    assert "no HIP device" in str(exc.value) or "device" in str(exc.value)
    G = np.zeros((6, 1))
    with pytest.raises(bioen_amd.BioenHipError):
        log_weights.bioen_log_posterior(np.zeros(6), G.copy(), G, yT, YT, 1.0, use_c=True)
    with pytest.raises(bioen_amd.BioenHipError):
        log_weights.find_optimum(G, G, yT, yT, YT, 1.0, minimize.Parameters("lbfgs"))
    w0 = np.full((6, 1), 1 / 6.0)
    with pytest.raises(bioen_amd.BioenHipError):
        forces.find_optimum(np.zeros((4, 1)), w0, yT, yT, YT, 1.0, minimize.Parameters("lbfgs"))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure; nothing under bioen_amd/ may reference it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bioen_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, fn)).read()
                if re.search(r"\boracle\b", txt) and "oracle" in txt.replace("the oracle", ""):
                    for line in txt.splitlines():
                        if re.search(r"(import|include|from)\s.*oracle", line):
                            bad.append((fn, line.strip()))
    assert not bad, bad


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("linesearch", [0, 1, 2, 3])
def test_lbfgs_driver_matches_oracle_driver_on_host_objectives(kind, linesearch):
    """The product's C++ L-BFGS driver + line searches (bioen_amd/csrc/lbfgs.{hpp,cpp}) run on
    a host objective must walk the same path as the oracle's C restatement of liblbfgs."""
    from bioen_amd import _lib
    from oracle import oracle_binding as O
    rng = np.random.default_rng(3)
    x0 = rng.uniform(-1.5, 1.5, 20)
    for cap in (1, 3, 8, 20):
        p = dict(LBFGS_DEFAULTS, linesearch=linesearch, epsilon=1e-12, delta=0.0, past=0, max_iterations=cap)
        x, info = _lib.selftest_lbfgs(kind, x0, p)
        xo, fo, co, ito, evo = O.selftest_lbfgs(kind, x0, p)
        assert (info.lbfgs_code, info.iterations, info.evaluations) == (co, ito, evo)
        assert np.abs(x - xo).max() < 1e-10
        assert abs(info.fmin - fo) <= 1e-9 * max(1.0, abs(fo))
    # and to convergence
    p = dict(LBFGS_DEFAULTS, linesearch=linesearch, epsilon=1e-9, delta=0.0, past=0, max_iterations=3000)
    x, info = _lib.selftest_lbfgs(kind, x0, p)
    assert info.lbfgs_code == 0 and np.abs(x - 1.0).max() < 1e-5


def test_lbfgs_driver_parameter_validation_order():
    from bioen_amd import _lib
    x0 = np.zeros(4)
    cases = [({"epsilon": -1.0}, -1017), ({"past": -1}, -1016), ({"delta": -1.0}, -1015), ({"ftol": -1.0}, -1011),
             ({"wolfe": 1.0}, -1010), ({"wolfe": 1e-6}, -1010), ({"gtol": -1.0}, -1009),
             ({"max_linesearch": 0}, -1007), ({"linesearch": 7}, -1014)]
    for mod, code in cases:
        _, info = _lib.selftest_lbfgs(0, x0, dict(LBFGS_DEFAULTS, **mod))
        assert info.lbfgs_code == code, (mod, info.lbfgs_code)
        assert info.evaluations == 0
    # x0 = minimiser of kind 1 -> "already minimized" (2)
    _, info = _lib.selftest_lbfgs(1, np.ones(5), LBFGS_DEFAULTS)
    assert info.lbfgs_code == 2


def _build_c_demo(tmp_path):
    """examples/c_abi_demo.c against include/bioen_hip.h and the shipped library, as strict C99"""
    import subprocess
    from bioen_amd import _lib
    exe = str(tmp_path / "c_abi_demo")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L", libdir, "-lbioen_hip",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe], check=True)
    return exe


def test_header_is_plain_c_and_a_c_caller_links(tmp_path):
    """the boundary is a C ABI: a C99 translation unit that includes the header and calls the path's entry points
    compiles without a warning and links against the library; without a GPU it reports that and leaves with 77 (no
    CPU path to fall back on)"""
    import subprocess
    exe = _build_c_demo(tmp_path)
    import bioen_amd
    if bioen_amd.device_count() == 0:
        p = subprocess.run([exe], capture_output=True, text=True)
        assert p.returncode == 77 and "no HIP device" in p.stdout


@pytest.mark.gpu
def test_c_caller_runs_the_hot_path(tmp_path):
    import subprocess
    p = subprocess.run([_build_c_demo(tmp_path)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (p.stdout, p.stderr)
