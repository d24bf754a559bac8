"""Forces strip passes at BASELINE config 5 size (N = 1e6 x M = 512 by default): per-launch time of the two
matrix passes at batch widths K = 1..8 through bioen_hip_forces_fdf_batch, and agreement of f / grad between
batch widths.  (An A/B against an earlier tree: build that tree beside this one and run its copy of this file.)"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bioen_amd

N = int(os.environ.get("FORCES_N", "1000000")); M = int(os.environ.get("FORCES_M", "512"))
reps = int(os.environ.get("REPS", "20"))
rng = np.random.default_rng(12345)
YTrue = rng.uniform(1, 10, M)
sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
YTilde = rng.normal(YTrue, sig_exp) / sig_exp
w0 = np.full(N, 1.0 / N)
thetas = np.logspace(3, -0.5, 8)
forces = 1e-3 * rng.standard_normal((8, M))
out = {"N": N, "M": M, "old": os.environ.get("BIOEN_HIP_STRIP_OLD", "0"), "K": {}}
with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
    f1, g1 = ctx.forces_fdf_batch(forces[:1], w0, thetas[:1])          # builds the strip copy, warms up
    ref = [ctx.forces_fdf_batch(forces[k:k + 1], w0, thetas[k:k + 1]) for k in range(8)]
    for K in [int(k) for k in os.environ.get("KS", "1,2,4,6,8").split(",")]:
        ctx.forces_fdf_batch(forces[:K], w0, thetas[:K])
        ctx.kernel_stats_enable(True); ctx.kernel_stats_reset(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f, g = ctx.forces_fdf_batch(forces[:K], w0, thetas[:K])
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / reps
        st = ctx.kernel_stats()
        same = all(f[k] == ref[k][0][0] and np.array_equal(g[k], ref[k][1][0]) for k in range(K))
        out["K"][K] = {"eval_ms": 1e3 * dt, "bt_ms": st["forward"]["total_ms"] / max(st["forward"]["launches"], 1),
                       "xy_ms": st["adjoint"]["total_ms"] / max(st["adjoint"]["launches"], 1),
                       "bitwise_equal_to_single": bool(same)}
        ctx.kernel_stats_enable(False)
    out["f"] = [float(r[0][0]) for r in ref]
print(json.dumps(out))

# ---- diagnostic build (-DSTRIP_DIAG=4): where a strip's cycles go, per phase, averaged over waves ------------------
if os.environ.get("STAMPS"):
    import ctypes as C
    from bioen_amd import _lib
    L = _lib.lib()
    fn = L.bioen_hip_debug_strip_stamps
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.c_int]
    names = ["wait+copy", "prefetch+P1", "barrier1", "P2", "barrier2", "P3", "-", "-"]
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        fn(ctx._h, 1, None, 0)
        for K in (1, 2, 3, 4, 8):
            res = {}
            for need_grad, tag in ((False, "xy"), (True, "bt")):
                ctx.forces_fdf_batch(forces[:K], w0, thetas[:K], need_grad=need_grad)
                ctx.forces_fdf_batch(forces[:K], w0, thetas[:K], need_grad=need_grad)
                nb = 256
                buf = (C.c_longlong * (nb * 16 * 8))()
                fn(ctx._h, 1, buf, nb)
                a = np.ctypeslib.as_array(buf).reshape(nb, 16, 8)[:, :8, :].astype(float)   # 8 waves per block
                strips = (N + 127) // 128 * 128 // 16 / nb
                res[tag] = {names[i]: round(a[:, :, i].mean() / strips) for i in range(6)}
                res[tag]["total"] = round(a[:, :, :6].sum(axis=2).mean() / strips)
            print("K=%d cycles per strip and wave:" % K, json.dumps(res))
