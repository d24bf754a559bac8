"""Development aid: PCIe-inclusive cost of handing a host matrix to bioen_hip_ctx_create."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
M, N = 1024, 1000000
t0 = time.perf_counter()
y = np.empty((M, N))
y[:] = 1.0
for i in range(0, M, 64):
    y[i:i + 64] += np.random.default_rng(i).standard_normal((min(64, M - i), N))
print("host matrix built in %.1f s" % (time.perf_counter() - t0)); sys.stdout.flush()
YT = np.zeros(M)
for rep in range(2):
    t0 = time.perf_counter()
    ctx = bioen_amd.Context(y, YT)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print("ctx_create upload of %.2f GB: %.3f s = %.1f GB/s" % (y.nbytes / 1e9, dt, y.nbytes / 1e9 / dt)); sys.stdout.flush()
    g = np.zeros(N)
    t0 = time.perf_counter()
    f, grad = ctx.logw_fdf(g, g, 10.0)
    print("first fdf incl. 3 N-vector transfers: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
    ctx.close()
