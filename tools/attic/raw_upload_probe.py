"""Input path at BASELINE config-3 size: yTilde = sim / sigma formed on the host (numpy)
and uploaded, against Context.from_raw (upload of the raw observables, division -- and for
structure-major input the transposition -- on the device)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bioen_amd

M, N = 1024, 1000000
err = np.linspace(0.1, 2.0, M)
exp = np.linspace(1.0, 10.0, M)
sim = np.empty((M, N))
sim[:] = np.linspace(1.0, 9.0, N)[None, :]
sim += np.arange(M)[:, None] * 1e-3
out = {}
t0 = time.perf_counter(); yTilde = sim / err[:, None]; YTilde = exp / err; out["host_division_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); ctx = bioen_amd.Context(yTilde, YTilde); ctx.synchronize(); out["upload_scaled_s"] = time.perf_counter() - t0
probe = ctx.read_ytilde(row0=3, rows=2, col0=999990, cols=10); ctx.close()
t0 = time.perf_counter(); ctx = bioen_amd.Context.from_raw(sim, exp, err); ctx.synchronize(); out["from_raw_s"] = time.perf_counter() - t0
assert np.array_equal(ctx.read_ytilde(row0=3, rows=2, col0=999990, cols=10), probe); ctx.close()
del yTilde
simT = np.empty((N, M))
simT[:] = (np.arange(M) * 1e-3)[None, :]
simT += np.linspace(1.0, 9.0, N)[:, None]
t0 = time.perf_counter(); hostT = np.ascontiguousarray(simT.T) / err[:, None]; out["host_transpose_division_s"] = time.perf_counter() - t0
del hostT
t0 = time.perf_counter(); ctx = bioen_amd.Context.from_raw(simT, exp, err, structure_major=True); ctx.synchronize()
out["from_raw_structure_major_s"] = time.perf_counter() - t0
assert np.allclose(ctx.read_ytilde(row0=3, rows=2, col0=999990, cols=10), probe, rtol=1e-15); ctx.close()
print(json.dumps(out))
