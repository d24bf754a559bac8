"""configs[4] (forces, N = 1e6 x M = 512, 8 thetas) with the engine's host-side timing on (BIOEN_HIP_FORCES_TIMING=1:
where the host's share of a round goes).  python tools/attic/forces_timing.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["BIOEN_HIP_FORCES_TIMING"] = "1"
from canon_probe import series         # noqa: E402

thetas = [float(t) for t in np.logspace(3, -0.5, 8)]
res, dt = series(512, 1000000, thetas, True, method="forces")
print("forces configs[4]: %.4f s, %d iterations, %d evaluations" % (dt, sum(r["iterations"] for r in res), sum(r["evaluations"] for r in res)))
