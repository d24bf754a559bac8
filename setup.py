"""Build hook of pyproject.toml: compile bioen_amd/libbioen_hip.so (hipcc, gfx950; bioen_amd/csrc/Makefile) before the
Python files are collected, so that the wheel carries the C-ABI library.  In-tree use needs none of this:
`make -C bioen_amd/csrc` (or `python -c 'import __graft_entry__ as e; e.build()'`) and `import bioen_amd`."""
import os
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py


class build_py_with_hip(build_py):
    def run(self):
        here = os.path.dirname(os.path.abspath(__file__))
        subprocess.check_call(["make", "-j", str(max(1, min(8, os.cpu_count() or 1))), "-C",
                               os.path.join(here, "bioen_amd", "csrc")])
        build_py.run(self)


setup(cmdclass={"build_py": build_py_with_hip})
