"""Build-container script: the time axis and the signal column of the DEER trace BASELINE.json configs[3] names
(/root/reference/examples/DEER/rotamer-refinement/single_trace/files/experimental_data/exp-370-292-signal-deer.dat,
205 points; column 0 = time in microseconds, column 2 = the signal BioEn fits: observables.py:340) as a small data
fixture -- numbers only; the reference does not travel to the GPU box.  usage: python tests/golden/make_deer_fixture.py"""
import os
import numpy as np

SRC = "/root/reference/examples/DEER/rotamer-refinement/single_trace/files/experimental_data/exp-370-292-signal-deer.dat"
a = np.loadtxt(SRC)
assert a.shape == (205, 3)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "deer_exp_370_292.npz")
np.savez_compressed(out, t_us=a[:, 0], raw=a[:, 1], signal=a[:, 2])
print("wrote", out, a.shape)
