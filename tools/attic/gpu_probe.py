"""Ad-hoc GPU probe (development aid): call latencies and small-problem iteration rates."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bioen_amd
from conftest import load_golden, LBFGS_DEFAULTS, LBFGS_TIGHT

d = load_golden("synth_logw_M64xN2000.npz")
t0 = time.perf_counter()
ctx = bioen_amd.Context(d["yTilde"], d["YTilde"])
print("ctx create %.3f s" % (time.perf_counter() - t0)); sys.stdout.flush()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.logw_fdf(d["GInit"], d["G"], d["theta"])
    print("logw_fdf: %.3f ms/call" % ((time.perf_counter() - t0) / 20 * 1e3)); sys.stdout.flush()
for params, tag in ((LBFGS_DEFAULTS, "def"), (LBFGS_TIGHT, "tight")):
    t0 = time.perf_counter()
    gopt, w, info = ctx.opt_lbfgs_logw(d["GInit"], d["G"], d["theta"], params)
    dt = time.perf_counter() - t0
    print("lbfgs %s: code %d, %d it, %d ev, %.3f s (info %.3f s) -> %.3f ms/iteration" % (
        tag, info.lbfgs_code, info.iterations, info.evaluations, dt, info.seconds, 1e3 * dt / max(info.iterations, 1)))
    sys.stdout.flush()
ctx.close()
