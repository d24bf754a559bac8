#!/usr/bin/env python3
"""The two log-weights matrix passes alone (bioen_hip_debug_pass_probe): mean time per launch of k_strip_fwd / the adjoint
kernel at batch widths K over the resident strip copies, HIP events.  A/B builds: BIOEN_HIP_LIBRARY=<other .so>.

    python3 tools/pass_probe.py [M] [N] [reps]          # GPU box
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
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rng = np.random.default_rng(12345)
YTrue = rng.uniform(1, 10, M)
sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
YTilde = rng.normal(YTrue, sig_exp) / sig_exp
out = {"M": M, "N": N, "reps": reps, "library": os.environ.get("BIOEN_HIP_LIBRARY", "default"),
       "one_copy": os.environ.get("BIOEN_HIP_ONE_COPY", "0"), "K": {}}
L = _lib.lib()
with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
    ctx.logw_fdf(np.zeros(N), np.zeros(N), 1.0)           # builds both strip copies
    bytes_ = ((M + 15) // 16 * 16) * ((N + 1023) // 1024 * 1024) * 8.0
    for rep in range(2):
        for K in [int(k) for k in os.environ.get("KS", "1,4,5,8").split(",")]:
            f, a = C.c_double(), C.c_double()
            _lib.check(L.bioen_hip_debug_pass_probe(ctx._h, K, reps, C.byref(f), C.byref(a)))
            out["K"].setdefault(K, []).append({"fwd_ms": round(f.value, 4), "adj_ms": round(a.value, 4),
                                               "fwd_TBs": round(bytes_ / f.value / 1e9, 3), "adj_TBs": round(bytes_ / a.value / 1e9, 3)})
    out["read_ceiling_GBs"] = ctx.read_probe(reps=10)
print(json.dumps(out))
