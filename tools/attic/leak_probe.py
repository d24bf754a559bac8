"""Development aid (GPU box): device and host memory before and after a few hundred context lifetimes (both methods, batches,
evaluations, storage formats, affine models): what a long-running caller would leak."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
from bench import LBFGS_DEFAULTS

hip = ctypes.CDLL("libamdhip64.so")


def free_bytes():
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value


def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1024.0


rng = np.random.default_rng(0)
M, N = 300, 20000
YTrue = rng.uniform(1, 10, M)
y = rng.normal(YTrue[:, None], 0.5 * YTrue[:, None], (M, N)) / (0.1 * YTrue[:, None])
YT = rng.normal(YTrue, 0.1 * YTrue) / (0.1 * YTrue)
G, w0, f0 = np.zeros(N), np.full(N, 1.0 / N), np.zeros(M)
params = dict(LBFGS_DEFAULTS, max_iterations=6)


def lifetime(k):
    with bioen_amd.Context(y, YT) as ctx:
        ctx.opt_lbfgs_logw_batch([100.0, 10.0, 1.0], G, G, params)
        ctx.opt_lbfgs_forces_batch([100.0, 10.0, 1.0, 0.3, 0.1], f0, w0, params, max_batch=5)
        ctx.logw_fdf(G, G, 1.0)
        if k % 3 == 0:
            ctx.set_storage("split")
            ctx.opt_lbfgs_logw(G, G, 10.0, params)
            ctx.set_storage("f64")
        if k % 4 == 0:
            ctx.set_affine(np.ones(M), np.full(M, 0.5))
            ctx.opt_lbfgs_logw(G, G, 10.0, params)
        ctx.opt_gsl_logw(G, G, 10.0, "bfgs2", dict(step_size=0.01, tol=0.001, max_iterations=3))


for k in range(5):
    lifetime(k)                       # warm: lazy allocations of the runtime itself
f0b, r0 = free_bytes(), rss_mb()
n = int(os.environ.get("LIFETIMES", "300"))
for k in range(n):
    lifetime(k)
f1b, r1 = free_bytes(), rss_mb()
print("%d context lifetimes: device memory free %.1f MB -> %.1f MB (delta %.2f MB), host RSS %.1f MB -> %.1f MB" % (
    n, f0b / 2 ** 20, f1b / 2 ** 20, (f1b - f0b) / 2 ** 20, r0, r1))
