#!/usr/bin/env python3
"""Where a strip copy lands in HBM and what that does to the rate it streams at (r06): contexts created one after the other in
ONE process -- kept or closed in between -- with the plain read probe over each copy (form 2 = row-sum order copy, the forward
pass's; form 4 = column-sum order copy, the adjoint's) and the two passes themselves at K = 4.

    python3 tools/placement_probe.py [M] [N] [contexts]          # GPU box
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np          # noqa: E402
import bioen_amd            # noqa: E402
from bioen_amd import _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 4
keep = os.environ.get("KEEP", "0") == "1"
rng = np.random.default_rng(12345)
YTrue = rng.uniform(1, 10, M)
sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
YTilde = rng.normal(YTrue, sig_exp) / sig_exp
L = _lib.lib()
alive = []
for i in range(count):
    ctx = bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345)
    ctx.logw_fdf(np.zeros(N), np.zeros(N), 1.0)
    row = {"context": i, "kept_alive_before": len(alive)}
    for rep in range(2):
        f, a = C.c_double(), C.c_double()
        _lib.check(L.bioen_hip_debug_pass_probe(ctx._h, 4, 20, C.byref(f), C.byref(a)))
        row.setdefault("fwd_ms", []).append(round(f.value, 4))
        row.setdefault("adj_ms", []).append(round(a.value, 4))
        row.setdefault("read_Ys_GBs", []).append(round(ctx.read_probe(reps=10, form=2)[0]))
        row.setdefault("read_Ys1_GBs", []).append(round(ctx.read_probe(reps=10, form=4)[0]))
    print(json.dumps(row), flush=True)
    if keep:
        alive.append(ctx)
    else:
        ctx.close()
for c in alive:
    c.close()
