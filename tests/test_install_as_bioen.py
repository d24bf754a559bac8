"""`bioen_amd.install_as_bioen()`: callers that say `from bioen import optimize` (bioen/analyze/procedure.py:9, the
reference's tests and notebooks) get this package without being edited.  Runs in a child interpreter so that the alias
does not leak into the other tests."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

CALLER = textwrap.dedent('''
    # a caller module as the reference writes them
    from bioen import optimize
    from bioen.optimize import log_weights, forces
    import bioen.optimize.ext.c_bioen as c_bioen

    def run():
        params = optimize.minimize.Parameters("lbfgs")
        return params, log_weights.find_optimum, forces.find_optimum, c_bioen.library_lbfgs()
''')


def test_unchanged_caller_imports_resolve_to_this_package(tmp_path):
    (tmp_path / "their_caller.py").write_text(CALLER)
    code = textwrap.dedent('''
        import sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import bioen_amd
        pkg = bioen_amd.install_as_bioen()
        import their_caller
        params, fo_logw, fo_forces, has_lbfgs = their_caller.run()
        assert their_caller.optimize is bioen_amd.optimize and pkg.optimize is bioen_amd.optimize
        assert fo_logw is bioen_amd.optimize.log_weights.find_optimum
        assert fo_forces is bioen_amd.optimize.forces.find_optimum
        assert their_caller.c_bioen is bioen_amd.optimize.ext.c_bioen and has_lbfgs is True
        assert params["minimizer"] in ("lbfgs", "liblbfgs") and "params" in params
        assert bioen_amd.install_as_bioen() is pkg          # idempotent
        print("alias ok")
    ''') % (ROOT, str(tmp_path))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path),
                         env=dict(os.environ, PYTHONPATH=""))
    assert out.returncode == 0 and "alias ok" in out.stdout, out.stderr


def test_packaging_metadata_names_the_library():
    import re
    with open(os.path.join(ROOT, "pyproject.toml")) as fp:
        text = fp.read()
    assert 'name = "bioen-amd"' in text and "libbioen_hip.so" in text
    import bioen_amd
    assert re.search(r'version = "%s"' % re.escape(bioen_amd.__version__), text)
