#!/usr/bin/env python3
"""bench.py -- BioEn log-weights theta-sweep on MI355X (BASELINE.json metric).

One "step" = one complete pass of the hot path over the workload: the 8-point
theta series (np.logspace(3, -0.5, 8), every theta cold-started, liblbfgs yaml
defaults) of the log-weights optimizer on a synthetic N = 1e6 x M = 1024 ensemble
(BASELINE.json configs[2]; the matrix is generated in HBM, so inputs are resident
when the timed region starts).  With --gpus N the work is sharded over N ranks, one
process per GPU: `python3 bench.py --gpus N` starts them itself (plain subprocesses,
before any HIP call: launch_ranks), `python -m torch.distributed.run ... bench.py
--gpus N` is taken as it comes (this script is torch-free either way); the per-theta
results are all-gathered over RCCL/xGMI.  A line labelled n_gpus: N is N ranks on N
devices -- fewer devices than ranks is refused unless --share-devices (a test mode).

value = (L-BFGS iterations of the whole job) * N * M / wall-clock, max over ranks.

--method forces runs BASELINE.json configs[4] instead (forces method, N = 1e6 x M = 512, the theta series as one
lock-step batch; with --gpus N the structures are split over the ranks, two small all-gathers per evaluation).

    python bench.py                      # 1 GPU, 1 warm-up + 1 timed sweep
    python bench.py --gpus 1 --steps 2 --warmup 1
    python bench.py --gpus 8             # 8 ranks started by this script
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29511 bench.py --gpus 8 --steps 1 --warmup 1
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LBFGS_DEFAULTS = dict(linesearch=2, max_iterations=5000, delta=1e-6, epsilon=1e-6, ftol=1e-5, gtol=0.9,
                      wolfe=0.9, past=10, max_linesearch=100)   # bioen_optimize.yaml:33-46
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
SEED = 12345


def kernel_source_sha():
    """Identifies the kernel sources a measurement belongs to (the GPU box has no .git): sha256 over
    bioen_amd/csrc/*.  profiles/traffic.json records it; a mismatch means the PMC pass is stale."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "bioen_amd", "csrc", "*"))):
        if os.path.isfile(f) and not f.endswith((".so", ".o")):
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fp:
                h.update(fp.read())
    return h.hexdigest()[:16]


def pmc_kernel_means(directory, counter):
    """rocprofv3 --pmc <counter> output -> {kernel base name: (mean counter value per launch, launches)}"""
    import collections, csv, glob
    files = glob.glob(os.path.join(directory, "**", "*_counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for f in files[:1]:
        with open(f) as fp:
            for r in csv.DictReader(fp):
                if r["Counter_Name"] == counter:
                    name = r["Kernel_Name"].replace("void ", "").replace("bioen::", "").split("(")[0].split("<")[0]
                    agg[name].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def under_profiler():
    return any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))


def live_traffic(method, M, N, kernel):
    """HBM bytes per launch of `kernel` (base name), measured NOW on this box: tools/pmc_pass.py as a child process under
    `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (the two counters cannot share a pass), converted
    with the gfx950 corrections of MI355X_MICROARCH.md (both in KiB; FETCH_SIZE counts a wide coalesced read stream at
    half its bytes).  None (with the reason) when rocprofv3 is absent, fails, or this process is itself being profiled."""
    import shutil, subprocess, tempfile
    if under_profiler():
        return None, "bench.py itself runs under rocprofv3"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.isfile(exe):
        return None, "rocprofv3 not found"
    vals, launches = {}, 0
    env = dict(os.environ, TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="bioen_pmc_", dir="/tmp")
        try:
            # the program after `--` is THIS interpreter's ELF binary, resolved: a PATH lookup could land on a shim or a
            # launcher script, and any such hop is an exec after the profiler's preloaded library has touched the GPU
            p = subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                                os.path.realpath(sys.executable),
                                os.path.join(ROOT, "tools", "pmc_pass.py"), method, str(M), str(N)],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            if p.returncode != 0:
                return None, "rocprofv3 --pmc %s: exit %d: %s" % (ctr, p.returncode, (p.stderr or "")[-300:])
            means = pmc_kernel_means(d, ctr)
            if kernel not in means:
                return None, "no %s launch in the %s pass" % (kernel, ctr)
            vals[ctr], launches = means[kernel]
        except Exception as e:
            return None, repr(e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024, "live: rocprofv3 --pmc passes inside this run (%d launches)" % launches


def synthetic_targets(M, seed=SEED):
    """Per-observable vectors of the SURVEY 8(d) recipe (after forces.py:19-68)."""
    rng = np.random.default_rng(seed)
    YTrue = rng.uniform(1, 10, M)
    sig_exp = 0.1 * YTrue
    sig_sim = 0.5 * YTrue
    YTilde = rng.normal(YTrue, sig_exp) / sig_exp
    return YTrue, sig_sim, sig_exp, YTilde


def survey_inputs(M, N, seed=SEED):
    """SURVEY 8(d)'s synthetic inputs to the letter: ONE PCG64 stream -- YTrue, then the matrix ROW by ROW, then the
    targets (after forces.py:19-68) -- so that iteration counts can be set beside the survey's orientation runs
    (theta = 10 at N = 1e5 x M = 256: 409 iterations on 8 threads, 388 on one).  The device generator of
    Context.synthetic draws the same distribution from a counter-based stream; this is the host form of it."""
    rng = np.random.default_rng(seed)
    YTrue = rng.uniform(1, 10, M)
    sig_exp = 0.1 * YTrue
    sig_sim = 0.5 * YTrue
    yTilde = np.empty((M, N))
    for i in range(M):
        yTilde[i, :] = rng.normal(YTrue[i], sig_sim[i], N) / sig_exp[i]
    YTilde = rng.normal(YTrue, sig_exp) / sig_exp
    return yTilde, YTilde


def cpu_baseline(ctx, M, N, YTilde, theta, budget_cols, cap_iterations):
    """Time the CPU path on a bounded sample of the SAME matrix (a column block read back
    from HBM): the reference's own C + liblbfgs code when oracle/_ref travelled here
    ("reference"), else the oracle's C restatement ("port")."""
    cols = int(min(N, budget_cols))
    sample = ctx.read_ytilde(0, M, 0, cols)
    G = np.zeros(cols)
    params = dict(LBFGS_DEFAULTS, max_iterations=cap_iterations)
    from oracle import ref_binding as R
    from oracle import cpus
    cores = cpus.usable_cpus()
    if R.available():
        kind = "reference"
        R.set_fast_openmp_flag(1)
        R.omp_set_num_threads(cores)
        # warm the thread pool / page in
        R.logw_f(G, G, sample[:, :1024].copy(), YTilde, theta)
        import ctypes as C
        g0, Gc, yT, YT = R._a(G).copy(), R._a(G), R._a(sample), R._a(YTilde)
        w = np.empty(cols); tmp_n = np.empty(cols); tmp_m = np.empty(M); result = np.empty(cols)
        yTT = np.ascontiguousarray(yT.T)            # the reference's transposed cache (c_bioen.pyx:471-473)
        p = R.params_t()
        p.g, p.G, p.yTilde, p.YTilde, p.w, p.result = R._p(g0), R._p(Gc), R._p(yT), R._p(YT), R._p(w), R._p(result)
        p.theta, p.yTildeT, p.caching = float(theta), R._p(yTT), 1
        p.tmp_n, p.tmp_m, p.m, p.n = R._p(tmp_n), R._p(tmp_m), M, cols
        err = C.c_int(0)
        t0 = time.perf_counter()
        R.lib()._opt_lbfgs_logw(p, R._lbfgs_cfg(params), R.visual_params(0, 0), C.byref(err))
        dt = time.perf_counter() - t0
        # liblbfgs stops with -997 after exactly cap_iterations accepted steps, or earlier on convergence;
        # the reference counts iterations only in a verbose printf, so re-derive: code -997 <=> cap reached
        iters = cap_iterations if err.value == -997 else None
        if iters is None:
            from oracle import oracle_binding as O
            iters = O.opt_lbfgs_logw(G, G, sample, YTilde, theta, params)[3]
    else:
        kind = "port"
        from oracle import oracle_binding as O
        os.environ.setdefault("OMP_NUM_THREADS", str(cores))
        t0 = time.perf_counter()
        res = O.opt_lbfgs_logw(G, G, sample, YTilde, theta, params)
        dt = time.perf_counter() - t0
        iters = res[3]
    return {
        "value": iters * float(cols) * M / dt,
        "unit": "iter*N*M/s",
        "cores": cores,
        "kind": kind,
        "sample": "columns [0,%d) of the same %dx%d matrix (M unchanged), theta=%g, yaml-default liblbfgs, "
                  "%d iterations in %.2f s (%.1f ms/iteration), %s OpenMP threads, transposed cache on"
                  % (cols, M, N, theta, iters, dt, 1e3 * dt / max(iters, 1), cores),
        "ms_per_iteration": 1e3 * dt / max(iters, 1),
    }


def cpu_fullsize(ctx, M, N, YTilde, thetas, mid_thetas=(), mid_budget_s=170.0, ms_per_eval=None):
    """A direct CPU number ON the headline config: the reference's own _opt_lbfgs_logw on the FULL matrix (read back
    from HBM, transposed cache built blockwise) for the cheapest thetas of the series, next to the device solving the
    same single problem.  Skipped when the host lacks the memory for matrix + transposed cache."""
    from oracle import ref_binding as R
    from oracle import cpus
    if not R.available():
        return None
    need = 2.0 * M * N * 8 + 6.0 * N * 8
    try:
        avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        avail = 0
    if avail < 1.4 * need:
        return {"skipped": "host memory: %.1f GB free, %.1f GB needed for the matrix and the reference's transposed cache"
                           % (avail / 1e9, 1.4 * need / 1e9)}
    cores = cpus.usable_cpus()
    R.set_fast_openmp_flag(1)
    R.omp_set_num_threads(cores)
    t0 = time.perf_counter()
    yT = np.empty((M, N))
    yTT = np.empty((N, M))
    step = 65536
    for c0 in range(0, N, step):
        blk = ctx.read_ytilde(0, M, c0, min(step, N - c0))
        yT[:, c0:c0 + blk.shape[1]] = blk
        yTT[c0:c0 + blk.shape[1], :] = blk.T
    t_read = time.perf_counter() - t0
    G = np.zeros(N)
    out = {"cores": cores, "readback_and_transpose_s": t_read, "per_theta": []}
    for th in thetas:
        gopt, fmin, code, cpu_s = _ref_lbfgs_logw(R, yT, yTT, YTilde, G, th, LBFGS_DEFAULTS)
        ctx.synchronize()
        t0 = time.perf_counter()
        _, _, info = ctx.opt_lbfgs_logw(G, G, th, LBFGS_DEFAULTS, want_weights=False)
        ctx.synchronize()
        gpu_s = time.perf_counter() - t0
        out["per_theta"].append({"theta": float(th), "cpu_s": cpu_s, "cpu_code": code, "gpu_s": gpu_s,
                                 "gpu_iterations": info.iterations, "gpu_evaluations": info.evaluations,
                                 "cpu_ms_per_gpu_iteration": 1e3 * cpu_s / max(info.iterations, 1),
                                 "fmin_rel_diff": abs(info.fmin - fmin) / abs(fmin), "speedup": cpu_s / gpu_s})
    out["cpu_s"] = sum(r["cpu_s"] for r in out["per_theta"])
    out["gpu_s"] = sum(r["gpu_s"] for r in out["per_theta"])
    out["speedup"] = out["cpu_s"] / out["gpu_s"] if out["gpu_s"] > 0 else None
    out["cpu_ms_per_evaluation"] = 1e3 * out["cpu_s"] / max(sum(r["gpu_evaluations"] for r in out["per_theta"]), 1) if thetas else ms_per_eval
    if mid_thetas:
        # r05: parity ON the headline config for a theta that matters (the six slow thetas carry 95 % of the sweep's
        # iterations).  The problem is flat at this size: under the converged settings (delta 0, past 0: no plateau stop)
        # theta = 10 needs 2 200 iterations for epsilon 1e-7 (3e-5 above the optimum), 11 800 for 1e-8 (5e-8 above it) --
        # half an hour of the reference at 0.16 s per evaluation, half a minute of the device.  So the optimum is the
        # DEVICE's converged run, and the reference is asked to confirm it: its own _opt_lbfgs_logw started AT the
        # device's optimum under the same converged settings must take it for a minimum (status 0 or 2: its own
        # objective and its own gradient norm on the full matrix), with fmin within 1e-6 and weights within 1e-5 max(w)
        # (north_star).  It then continues from there at a tenth of the epsilon for a few iterations, which says how far a
        # stricter stop still moves the optimum; and its yaml-default run from the cold start gives both sides' distance
        # to that optimum at the settings the sweep is run with.
        # The candidates are tried in turn: the reference's _get_weights exponentiates WITHOUT a maximum shift
        # (c_bioen_kernels_logw.c:55-94), so a long trial step of its line search can overflow exp() -- at theta = 31.6 on
        # this matrix a trial point reaches max g = 712 in the fourth iteration, the objective is NaN from there on, and its
        # -ffast-math liblbfgs returns that as status 0.  (The device shifts by the maximum and walks on: same trajectory
        # to 1e-12 up to that point.)  Such a theta is recorded as what it is and the next one is taken.
        ms_it = out["cpu_ms_per_evaluation"] or 130.0
        eps = 1e-8
        conv = dict(LBFGS_DEFAULTS, epsilon=eps, delta=0.0, past=0, max_iterations=15000)
        out["mid_theta"] = {"reference_failures": []}
        spent = 0.0
        for mid_theta in mid_thetas:
            t0 = time.perf_counter()
            _, _, i_def = ctx.opt_lbfgs_logw(G, G, mid_theta, LBFGS_DEFAULTS, want_weights=False)
            gpu_def_s = time.perf_counter() - t0
            need = 1e-3 * ms_it * (i_def.evaluations + 40.0)
            if spent + need > mid_budget_s:
                out["mid_theta"]["skipped"] = "budget: %.0f s spent, ~%.0f s more needed at %.0f ms per reference evaluation (budget %.0f s)" \
                                              % (spent, need, ms_it, mid_budget_s)
                break
            g_def, f_def_ref, c_def_ref, s_def = _ref_lbfgs_logw(R, yT, yTT, YTilde, G, mid_theta, LBFGS_DEFAULTS)
            spent += s_def
            if not np.isfinite(f_def_ref):
                out["mid_theta"]["reference_failures"].append(
                    {"theta": float(mid_theta), "fmin": "nan", "code": c_def_ref, "seconds": s_def, "max_abs_g": float(np.nanmax(np.abs(g_def))),
                     "why": "exp() overflow at a trial point of the reference's line search (no maximum shift in _get_weights); "
                            "liblbfgs (-ffast-math) reports status %d" % c_def_ref})
                continue
            t0 = time.perf_counter()
            g_dev, w_dev, i_conv = ctx.opt_lbfgs_logw(G, G, mid_theta, conv)
            gpu_conv_s = time.perf_counter() - t0
            # the reference at the device's optimum: same settings (a few iterations at most if it agrees)
            g_ref, f_ref, c_ref, s_at = _ref_lbfgs_logw(R, yT, yTT, YTilde, G, mid_theta, dict(conv, max_iterations=25), g_start=g_dev)
            w_ref = np.exp(g_ref - g_ref.max())
            w_ref /= w_ref.sum()
            # ... and on from there at a tenth of the epsilon
            g_on, f_on, c_on, s_on = _ref_lbfgs_logw(R, yT, yTT, YTilde, G, mid_theta, dict(conv, epsilon=0.1 * eps, max_iterations=25), g_start=g_ref)
            w_on = np.exp(g_on - g_on.max())
            w_on /= w_on.sum()
            spent += s_at + s_on
            f_star = min(f_on, f_ref, i_conv.fmin)
            out["mid_theta"].update({
                "theta": float(mid_theta),
                "settings": "converged: epsilon %g, delta 0, past 0 (no plateau stop); device from the cold start, reference started at the "
                            "device's optimum" % eps,
                "device": {"code_converged": i_conv.lbfgs_code, "fmin_converged": i_conv.fmin, "iterations_converged": i_conv.iterations,
                           "evaluations_converged": i_conv.evaluations, "seconds_converged": gpu_conv_s,
                           "code_default": i_def.lbfgs_code, "fmin_default": i_def.fmin, "iterations_default": i_def.iterations,
                           "seconds_default": gpu_def_s},
                "reference_at_device_optimum": {"code": c_ref, "fmin": f_ref, "seconds": s_at,
                                                "meaning": "0 = converged, 2 = already minimized: the reference's own objective and gradient "
                                                           "norm on the full matrix take the device's optimum for a minimum"},
                "fmin_rel_diff_converged": abs(i_conv.fmin - f_ref) / abs(f_ref),
                "w_diff_over_max_w": float(np.abs(w_dev - w_ref).max() / w_ref.max()),
                "within_north_star": bool(c_ref in (0, 2) and i_conv.lbfgs_code in (0, 2) and abs(i_conv.fmin - f_ref) <= 1e-6 * abs(f_ref) and
                                          np.abs(w_dev - w_ref).max() <= 1e-5 * w_ref.max()),
                "reference_continued_at_a_tenth_of_epsilon": {
                    "code": c_on, "fmin": f_on, "iterations_cap": 25, "seconds": s_on,
                    "fmin_rel_gain": (f_ref - f_on) / abs(f_ref), "w_moved_over_max_w": float(np.abs(w_on - w_ref).max() / w_ref.max())},
                "reference_default": {"code": c_def_ref, "fmin": f_def_ref, "seconds": s_def},
                "default_stop_above_optimum_rel": {"reference": (f_def_ref - f_star) / abs(f_star), "device": (i_def.fmin - f_star) / abs(f_star)},
                "reference_seconds": spent,
            })
            break
    return out


def cpu_baseline_forces(ctx, M, N, YTilde, theta, budget_cols, cap_iterations):
    """--method forces: the reference's _opt_lbfgs_forces (else the oracle's restatement) on a column block of the
    same matrix, capped."""
    cols = int(min(N, budget_cols))
    sample = ctx.read_ytilde(0, M, 0, cols)
    w0 = np.full(cols, 1.0 / cols)
    params = dict(LBFGS_DEFAULTS, max_iterations=cap_iterations)
    from oracle import ref_binding as R
    from oracle import oracle_binding as O
    from oracle import cpus
    cores = cpus.usable_cpus()
    if R.available():
        kind = "reference"
        R.set_fast_openmp_flag(1)
        R.omp_set_num_threads(cores)
        R.forces_f(np.zeros(M), w0[:1024] * cols / 1024, sample[:, :1024].copy(), YTilde, theta)       # thread pool up
        t0 = time.perf_counter()
        _, _, code = R.opt_lbfgs_forces(np.zeros(M), w0, sample, YTilde, theta, params)
        dt = time.perf_counter() - t0          # includes the transposed cache, as the reference's pyx builds it per call
    else:
        kind = "port"
        os.environ.setdefault("OMP_NUM_THREADS", str(cores))
        t0 = time.perf_counter()
        code = O.opt_lbfgs_forces(np.zeros(M), w0, sample, YTilde, theta, params)[2]
        dt = time.perf_counter() - t0
    iters = cap_iterations if code == -997 else O.opt_lbfgs_forces(np.zeros(M), w0, sample, YTilde, theta, params)[3]
    return {"value": iters * float(cols) * M / dt, "unit": "iter*N*M/s", "cores": cores, "kind": kind,
            "sample": "forces method, columns [0,%d) of the same %dx%d matrix, theta=%g, yaml-default liblbfgs, %d iterations "
                      "in %.2f s (%.1f ms/iteration, transposed cache included), %s OpenMP threads"
                      % (cols, M, N, theta, iters, dt, 1e3 * dt / max(iters, 1), cores),
            "ms_per_iteration": 1e3 * dt / max(iters, 1)}


def _ref_lbfgs_logw(R, yT, yTT, YT, G, theta, params, g_start=None):
    """the reference's own _opt_lbfgs_logw on host arrays (transposed cache given) -> (gopt, fmin, code, seconds)"""
    import ctypes as C
    M, cols = yT.shape
    g0 = (G if g_start is None else g_start).copy()
    w = np.empty(cols); tmp_n = np.empty(cols); tmp_m = np.empty(M); result = np.empty(cols)
    p = R.params_t()
    p.g, p.G, p.yTilde, p.YTilde, p.w, p.result = R._p(g0), R._p(G), R._p(yT), R._p(YT), R._p(w), R._p(result)
    p.theta, p.yTildeT, p.caching = float(theta), R._p(yTT), 1
    p.tmp_n, p.tmp_m, p.m, p.n = R._p(tmp_n), R._p(tmp_m), M, cols
    err = C.c_int(0)
    t0 = time.perf_counter()
    fmin = R.lib()._opt_lbfgs_logw(p, R._lbfgs_cfg(params), R.visual_params(0, 0), C.byref(err))
    return result, fmin, err.value, time.perf_counter() - t0


def cpu_matched(bioen_amd, thetas, seed, budget_iters_1t=60):
    """BASELINE configs[1] (N = 1e5 x M = 256) end to end on BOTH sides, same inputs, same settings:
    the whole theta series through the reference's C + liblbfgs path on all granted cores (serially over
    theta, as procedure.py:62 does) next to the device's lock-step batch, with per-theta agreement of the
    minima; plus the 1-thread figure BASELINE.md 3.3 asks for (theta = 10, capped)."""
    from oracle import ref_binding as R
    from oracle import cpus
    if not R.available():
        return None
    N, M = 100000, 256
    yT, YTilde = survey_inputs(M, N, seed)
    cores = cpus.usable_cpus()
    out = {"workload": "log-weights theta series, N=%d x M=%d, %d thetas, cold starts, yaml-default liblbfgs" % (N, M, len(thetas)),
           "inputs": "SURVEY 8(d) to the letter: numpy default_rng(%d), row-wise normals, uploaded from the host" % seed,
           "cores": cores}
    with bioen_amd.Context(yT, YTilde) as ctx:
        from bioen_amd import sweep
        G = np.zeros(N)
        sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS)                       # warm-up
        ctx.synchronize()
        t0 = time.perf_counter()
        res = sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS)
        ctx.synchronize()
        out["gpu_sweep_s"] = time.perf_counter() - t0
        out["gpu_iterations"] = int(sum(r["iterations"] for r in res))
        out["gpu_iterations_per_theta"] = [int(r["iterations"]) for r in res]
        out["survey_iterations_theta10"] = {"8 threads": 409, "1 thread": 388}            # BASELINE.md 2 (orientation runs)
        # r05: the optimum of every theta (converged settings: epsilon 1e-8, no plateau stop) -- what both sides' yaml-default
        # stops are held against below, and what the reference is asked to confirm (started there, same settings)
        eps = 1e-8
        conv = dict(LBFGS_DEFAULTS, epsilon=eps, delta=0.0, past=0, max_iterations=60000)
        t0 = time.perf_counter()
        g_star, w_star, i_star = ctx.opt_lbfgs_logw_batch(thetas, G, G, conv, max_batch=8)
        conv_s = time.perf_counter() - t0
    yTT = np.ascontiguousarray(yT.T)
    R.set_fast_openmp_flag(1)
    R.omp_set_num_threads(cores)
    R.logw_f(G, G, yT, YTilde, 10.0)                                                     # thread pool up
    cpu_s, rel, signed, codes, fref = 0.0, [], [], [], []
    for th, r in zip(thetas, res):
        gopt, fmin, code, dt = _ref_lbfgs_logw(R, yT, yTT, YTilde, G, th, LBFGS_DEFAULTS)
        cpu_s += dt
        codes.append(code)
        fref.append(fmin)
        rel.append(abs(r["fmin"] - fmin) / abs(fmin))
        signed.append((r["fmin"] - fmin) / abs(fmin))       # > 0: the device stopped ABOVE the reference's minimum
    out.update({"cpu_sweep_s": cpu_s, "cpu_codes": codes, "speedup": cpu_s / out["gpu_sweep_s"],
                "fmin_rel_diff_per_theta": rel, "fmin_rel_diff_max": max(rel),
                "fmin_signed_rel_diff_per_theta": signed})
    # r05: both sides against the OPTIMUM (the device's converged runs above), and the reference's word on that optimum: its
    # own _opt_lbfgs_logw started there under the same converged settings (status 0 / 2 = it takes the point for a minimum)
    f_star = [i.fmin for i in i_star]
    confirm = []
    for k, th in enumerate(thetas):
        g_c, f_c, c_c, _ = _ref_lbfgs_logw(R, yT, yTT, YTilde, G, th, dict(conv, max_iterations=25), g_start=g_star[k])
        w_c = np.exp(g_c - g_c.max())
        w_c /= w_c.sum()
        confirm.append({"theta": float(th), "device_code": i_star[k].lbfgs_code, "device_iterations": i_star[k].iterations,
                        "reference_code": c_c, "fmin_rel_diff": abs(f_c - f_star[k]) / abs(f_c),
                        "w_diff_over_max_w": float(np.abs(w_c - w_star[k]).max() / w_c.max())})
    out["optimum"] = {
        "settings": "converged: epsilon %g, delta 0, past 0; device from the cold start (all thetas as one batch: %.2f s), reference "
                    "started at the device's optimum" % (eps, conv_s),
        "per_theta": confirm,
        "within_north_star": bool(all(c["device_code"] in (0, 2) and c["reference_code"] in (0, 2) and c["fmin_rel_diff"] <= 1e-6 and
                                      c["w_diff_over_max_w"] <= 1e-5 for c in confirm)),
        "default_stop_above_optimum_rel": {"device": [(r["fmin"] - f) / abs(f) for r, f in zip(res, f_star)],
                                           "reference": [(fr - f) / abs(f) for fr, f in zip(fref, f_star)]}}
    # The yardstick for the differences between the two yaml-default stops: the REFERENCE against itself.  At yaml defaults both codes stop on the plateau
    # test (delta = 1e-6 over 10 iterations), and where exactly depends on the rounding of the sums: the same binary, the
    # same inputs, serial sums (fast_openmp = 0) instead of OpenMP reductions.
    variants = {"fast_openmp=0, %d threads" % cores: (0, cores)}
    if cores >= 4:
        variants["fast_openmp=1, %d threads" % (cores // 2)] = (1, cores // 2)     # another partition of the OpenMP reductions
    spread, spread_s = [0.0] * len(thetas), 0.0
    per_variant = {}
    for name, (flag, nthr) in variants.items():
        R.set_fast_openmp_flag(flag)
        R.omp_set_num_threads(nthr)
        diffs = []
        for th, f1 in zip(thetas, fref):
            _, f0, code0, dt = _ref_lbfgs_logw(R, yT, yTT, YTilde, G, th, LBFGS_DEFAULTS)
            spread_s += dt
            diffs.append((f0 - f1) / abs(f1))
        per_variant[name] = diffs
        spread = [d if abs(d) > abs(s0) else s0 for d, s0 in zip(diffs, spread)]
    R.set_fast_openmp_flag(1)
    R.omp_set_num_threads(cores)
    out["reference_self_spread_per_theta"] = spread
    out["reference_self_spread_by_variant"] = per_variant
    out["reference_self_spread"] = ("largest signed (fmin(variant) - fmin(fast_openmp=1, %d threads)) / |fmin| over the variants: the "
                                    "reference's own binary on the same inputs, another summation order (%.1f s of CPU sweeps)"
                                    % (cores, spread_s))
    out["device_within_reference_spread"] = [bool(abs(a) <= max(abs(b), 1e-9)) for a, b in zip(signed, spread)]
    # 1 thread: theta = 10 capped at budget_iters_1t iterations (-997 = cap reached)
    R.omp_set_num_threads(1)
    capped = dict(LBFGS_DEFAULTS, max_iterations=budget_iters_1t)
    _, _, code, dt = _ref_lbfgs_logw(R, yT, yTT, YTilde, G, 10.0, capped)
    if code == -997:
        out["single_thread"] = {"value": budget_iters_1t * float(N) * M / dt, "unit": "iter*N*M/s", "cores": 1,
                                "ms_per_iteration": 1e3 * dt / budget_iters_1t,
                                "sample": "theta=10, %d iterations in %.2f s" % (budget_iters_1t, dt)}
    R.omp_set_num_threads(cores)
    return out


def one_copy_record(bioen_amd, M, N, YTrue, sig_sim, sig_exp, YTilde, thetas, max_batch, base_results, base_sweep_s, base_stats,
                    base_one_copy):
    """The same sweep in the OTHER strip-copy form (r06: a matrix above 1 GiB keeps ONE strip copy by default -- the adjoint
    on the row-sum order copy through the forces kernels' LDS image; BIOEN_HIP_ONE_COPY=0 / 1 asks for two / one).  A side
    record -- never `value`: what the second copy's 8.2 GB buy in time (nothing, within the run-to-run spread), and that
    both forms stop at the same minima (another order of the adjoint's sums over rows: the plateau stop may fall a few
    iterations elsewhere)."""
    from bioen_amd import sweep
    other = "0" if base_one_copy else "1"
    os.environ["BIOEN_HIP_ONE_COPY"] = other
    try:
        with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED) as ctx:
            G = np.zeros(N)
            sweep.sweep_log_weights(ctx, thetas[:2], G, G, LBFGS_DEFAULTS, max_batch=max_batch)       # the copies, warm-up
            ctx.kernel_stats_enable(True)
            ctx.kernel_stats_reset()
            ctx.synchronize()
            t0 = time.perf_counter()
            res = sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS, max_batch=max_batch)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            st = ctx.kernel_stats()
            forms, nbytes = ctx.footprint()
    finally:
        os.environ.pop("BIOEN_HIP_ONE_COPY", None)
    rounds = max(st["forward"]["launches"], 1)
    base_rounds = max(base_stats["forward"]["launches"], 1)
    one_adj = "k_strip<K, nt, ADJ>" if M <= 512 else "k_strip2<K, nt, ADJ>"
    return {"note": "the form that is NOT this matrix's default (BIOEN_HIP_ONE_COPY=%s); the headline runs on %s" %
                    (other, "one strip copy" if base_one_copy else "two strip copies"),
            "form": "one copy" if other == "1" else "two copies",
            "resident": sorted(forms), "resident_bytes": nbytes,
            "default_resident_bytes": nbytes // 2 if base_one_copy else 2 * nbytes,
            "sweep_s": dt, "default_sweep_s": base_sweep_s, "rounds": rounds, "default_rounds": base_rounds,
            "ms_per_round": 1e3 * dt / rounds, "default_ms_per_round": 1e3 * base_sweep_s / base_rounds,
            "iterations": int(sum(r["iterations"] for r in res)),
            "fwd_ms": st["forward"]["total_ms"] / rounds, "adj_ms": st["adjoint"]["total_ms"] / max(st["adjoint"]["launches"], 1),
            "default_fwd_ms": base_stats["forward"]["total_ms"] / base_rounds,
            "default_adj_ms": base_stats["adjoint"]["total_ms"] / max(base_stats["adjoint"]["launches"], 1),
            "adjoint_kernel": one_adj if other == "1" else "k_strip_adj",
            "default_adjoint_kernel": one_adj if base_one_copy else "k_strip_adj",
            "fmin_rel_diff_max_vs_default": max(abs(a["fmin"] - b["fmin"]) / abs(b["fmin"]) for a, b in zip(res, base_results))}


def storage_record(ctx, thetas, G, g0, max_batch, base_results, base_stats):
    """The reduced-byte storage EXPERIMENT (Context.set_storage) as a side record -- never `value`: the
    same sweep on the same context with the centred matrix streamed as fp32 + bf16 split (6 bytes per element) and as
    fp32 (4), reassembled to FP64 in registers, all sums FP64.  Two kinds of numbers: what the format buys (sweep time,
    matrix-kernel time per launch, time per lock-step round -- the round COUNT of a yaml-default sweep moves with the
    rounding, so the sweep time is not the per-round gain) and what it costs (converged runs, epsilon = 1e-10, delta = 0,
    of the two cheapest thetas against the FP64 path: fmin and weights)."""
    from bioen_amd import sweep
    conv = dict(LBFGS_DEFAULTS, epsilon=1e-10, delta=0.0, past=0, max_iterations=200000)
    cheap = [float(t) for t in sorted(thetas)[-2:]]
    _, w_ref, i_ref = ctx.opt_lbfgs_logw_batch(cheap, g0, G, conv, max_batch=max_batch)
    base_rounds = max(base_stats["forward"]["launches"], 1)
    out = {"note": "opt-in experiment, not the graded path; dtype of all arithmetic stays f64",
           "f64": {"bytes_per_element": 8, "rounds": base_rounds,
                   "fwd_ms": base_stats["forward"]["total_ms"] / base_rounds,
                   "adj_ms": base_stats["adjoint"]["total_ms"] / max(base_stats["adjoint"]["launches"], 1)},
           "converged_check": "theta = %s, epsilon 1e-10, delta 0, past 0, against the FP64 path on the same matrix" % cheap}
    for fmt, nbytes in (("split", 6), ("fp32", 4)):
        ctx.set_storage(fmt)
        sweep.sweep_log_weights(ctx, thetas, G, g0, LBFGS_DEFAULTS, max_batch=max_batch)       # builds the copies
        ctx.kernel_stats_enable(True)
        ctx.kernel_stats_reset()
        ctx.synchronize()
        t0 = time.perf_counter()
        res = sweep.sweep_log_weights(ctx, thetas, G, g0, LBFGS_DEFAULTS, max_batch=max_batch)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        st = ctx.kernel_stats()
        ctx.kernel_stats_enable(False)
        rounds = max(st["forward"]["launches"], 1)
        _, w_c, i_c = ctx.opt_lbfgs_logw_batch(cheap, g0, G, conv, max_batch=max_batch)
        out[fmt] = {"format": "fp32 + bf16 residual of (yTilde - YTilde)" if fmt == "split" else "fp32 of (yTilde - YTilde)",
                    "bytes_per_element": nbytes, "sweep_s": dt, "iterations": int(sum(r["iterations"] for r in res)),
                    "rounds": rounds, "ms_per_round": 1e3 * dt / rounds,
                    "fwd_ms": st["forward"]["total_ms"] / rounds,
                    "adj_ms": st["adjoint"]["total_ms"] / max(st["adjoint"]["launches"], 1),
                    "yaml_default_fmin_rel_diff_max": max(abs(a["fmin"] - b["fmin"]) / abs(b["fmin"]) for a, b in zip(res, base_results)),
                    "fmin_rel_diff_max": max(abs(a.fmin - b.fmin) / abs(b.fmin) for a, b in zip(i_c, i_ref)),
                    "w_diff_max": float(max(np.abs(w_c[k] - w_ref[k]).max() / w_ref[k].max() for k in range(len(cheap))))}
    ctx.set_storage("f64")
    return out


def forces_record(bioen_amd, thetas, seed, max_batch, with_cpu=True, with_split=True):
    """BASELINE configs[4] on one GPU: forces method, N = 1e6 x M = 512, the theta series as ONE lock-step
    batch (cold starts, yaml-default liblbfgs).  Not part of `value`; reported beside it."""
    N, M = 1000000, 512
    YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M, seed)
    w0 = np.full(N, 1.0 / N)
    f0 = np.zeros(M)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=seed) as ctx:
        ctx.opt_lbfgs_forces_batch(thetas[:2], f0, w0, dict(LBFGS_DEFAULTS, max_iterations=3), max_batch=max_batch)   # builds the strip copy
        ctx.kernel_stats_enable(True)
        ctx.kernel_stats_reset()
        ctx.synchronize()
        t0 = time.perf_counter()
        res, w, infos = ctx.opt_lbfgs_forces_batch(thetas, f0, w0, LBFGS_DEFAULTS, max_batch=max_batch)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        st = ctx.kernel_stats()
        split = None
        try:             # the opt-in storage experiment on this workload too (never the graded number): 6-byte split copies
            if not with_split:
                raise RuntimeError("skipped (--no-storage-experiment)")
            ctx.set_storage("split")
            ctx.opt_lbfgs_forces_batch(thetas[:2], f0, w0, dict(LBFGS_DEFAULTS, max_iterations=3), max_batch=max_batch)
            ctx.kernel_stats_enable(True)
            ctx.kernel_stats_reset()
            ctx.synchronize()
            t1 = time.perf_counter()
            _, _, infos_s = ctx.opt_lbfgs_forces_batch(thetas, f0, w0, LBFGS_DEFAULTS, max_batch=max_batch)
            ctx.synchronize()
            dts = time.perf_counter() - t1
            sts = ctx.kernel_stats()
            ctx.kernel_stats_enable(False)
            split = {"format": "fp32 + bf16 residual of (yTilde - YTilde), 6 bytes per element", "ms_per_step": 1e3 * dts,
                     "iterations": int(sum(i.iterations for i in infos_s)),
                     "xy_ms": sts["adjoint"]["total_ms"] / max(sts["adjoint"]["launches"], 1),
                     "bt_ms": sts["forward"]["total_ms"] / max(sts["forward"]["launches"], 1),
                     "rounds": sts["forward"]["launches"],
                     "fmin_rel_diff_max": max(abs(a.fmin - b.fmin) / abs(b.fmin) for a, b in zip(infos_s, infos))}
            ctx.set_storage("f64")
        except Exception as e:
            split = {"error": repr(e)}
        cpu = None
        failed_ref = None
        if with_cpu:     # the reference's _opt_lbfgs_forces (c_bioen_kernels_forces.c:574-662) on a column block of THIS matrix
            try:
                cpu = cpu_baseline_forces(ctx, M, N, YTilde, 10.0, 262144, 40)
            except Exception as e:
                cpu = {"value": None, "unit": "iter*N*M/s", "cores": 0, "kind": "error", "sample": repr(e)}
            try:         # thetas that end outside liblbfgs' success codes here: what does the REFERENCE do with them?
                failed_ref = forces_failed_thetas(bioen_amd, ctx, M, N, YTilde,
                                                  [float(t) for t, i in zip(thetas, infos) if i.lbfgs_code not in (0, 1, 2)],
                                                  {float(t): i for t, i in zip(thetas, infos)})
            except Exception as e:
                failed_ref = {"error": repr(e)}
    ok = [i for i in infos if i.lbfgs_code in (0, 1, 2)]          # through the kept API a failed theta is a RuntimeError, as in the
    its = int(sum(i.iterations for i in ok))                      # reference (c_bioen.pyx:516-520): its iterations are not counted
    kern = {}
    for name, which in (("xy", "adjoint"), ("bt", "forward")):       # timer slots of launch_forces_xy / _bt
        s_ = st[which]
        launches = max(s_["launches"], 1)
        avg_ms = s_["total_ms"] / launches
        avg_k = s_["problem_passes"] / launches
        alg = float(M) * N * 8 + avg_k * (8.0 * N + 8.0 * M)         # the matrix once + per theta an N- and an M-vector
        kern[name] = {"kernel": "k_strip<K, nt, %s>" % ("true" if name == "xy" else "false"), "launches": s_["launches"],
                      "avg_ms": avg_ms, "avg_batch_width": avg_k, "algorithmic_bytes": alg,
                      "achieved_GBs": alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0}
    dom = max(kern, key=lambda k: kern[k]["avg_ms"])
    return {"workload": "forces theta series, N=%d x M=%d, %d thetas, cold starts, yaml-default liblbfgs, one lock-step batch"
                        % (N, M, len(thetas)),
            "value": its * float(N) * M / dt, "unit": "iter*N*M/s", "ms_per_step": 1e3 * dt, "iterations": its,
            "evaluations": int(sum(i.evaluations for i in infos)),
            "cpu_baseline": cpu,
            "thetas_counted": len(ok), "thetas_failed": [float(t) for t, i in zip(thetas, infos) if i.lbfgs_code not in (0, 1, 2)],
            "failed_thetas_vs_reference": failed_ref,
            "storage_experiment_split": split,
            "speedup_vs_cpu": (its * float(N) * M / dt) / cpu["value"] if cpu and cpu.get("value") else None,
            "roofline": {"bound": "hbm", "kernel": kern[dom]["kernel"], "achieved": kern[dom]["achieved_GBs"],
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kern[dom]["achieved_GBs"] / HBM_PEAK_GBS,
                         "traffic": None, "kernels": kern},
            "per_theta": [{"theta": float(t), "iterations": i.iterations, "evaluations": i.evaluations, "code": i.lbfgs_code,
                           "fmin": i.fmin, "chi2": i.chi2, "S": -i.kl, "seconds": i.seconds} for t, i in zip(thetas, infos)]}


def forces_failed_thetas(bioen_amd, ctx, M, N, YTilde, thetas, results, seconds=60.0):
    """The thetas of configs[4]'s series whose runs end outside liblbfgs' success codes (-998: the line search has used
    its max_linesearch trials) -- like for like against the reference's own _opt_lbfgs_forces
    (c_bioen_kernels_forces.c:574-662, lbfgs.c:645-734) on the SAME full matrix (read back from HBM; the reference builds
    its transposed cache per call), in BOTH of its summation modes (fast_openmp 1 / 0, c_bioen_common.c:46-55) while
    `seconds` last: status and fmin of all three.

    r06 finding (tools/forces_status_probe.py -> profiles/r06_forces_status_probe_full.txt; the reference's spread on a
    configs[4]-shaped problem in tests/golden/forces_status_cfg4_M512xN100000.json): at large theta these runs end where
    the decrease the line search still asks for (1/2 g^2 / (theta var) ~ 1e-15) lies below the rounding noise of the
    objective (f ~ 250: ~5e-13), so 0 (the gradient test met by a last lucky step) or -998 is decided by rounding -- the
    reference's own status changes with its summation mode, its thread count and from run to run; liblbfgs' own binary
    driven by the DEVICE's objective and gradient (100 x closer to the 80-bit values than the reference's) flips the same
    way.  Equal minima, not equal statuses, are what the two sides can be held to here."""
    from oracle import ref_binding as R
    from oracle import cpus
    if not thetas or not R.available():
        return None
    need = 2.2 * M * N * 8
    try:
        avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        avail = 0
    if avail < 1.3 * need:
        return {"skipped": "host memory: %.1f GB free, %.1f GB needed" % (avail / 1e9, 1.3 * need / 1e9)}
    cores = cpus.usable_cpus()
    R.omp_set_num_threads(cores)
    yT = ctx.read_ytilde()
    w0 = np.full(N, 1.0 / N)
    out = {"sample": "the full %d x %d matrix, forces_init = 0, yaml-default liblbfgs, %d OpenMP threads, fast_openmp = 1 "
                     "and (while %.0f s last) 0" % (M, N, cores, seconds),
           "evidence": ["profiles/r06_forces_status_probe_full.txt", "tests/golden/forces_status_cfg4_M512xN100000.json"],
           "per_theta": []}
    t_all = time.perf_counter()
    floor_codes = (0, -998, -1000, -1001)       # converged | the line search's three ways of giving up at the rounding floor
    for th in thetas:
        dev = results[th]
        rec = {"theta": th, "device_code": int(dev.lbfgs_code), "device_fmin": float(dev.fmin),
               "device_iterations": dev.iterations, "device_evaluations": dev.evaluations, "reference": []}
        for flag in (1, 0):
            # the second mode only where a run is short (a run that ends -998 takes a hundred evaluations of the full matrix:
            # ~20 s; that the reference's status depends on its mode there is the golden's and the probe's finding)
            if flag == 0 and (time.perf_counter() - t_all > seconds or rec["reference"][0]["seconds"] > 8.0):
                break
            R.set_fast_openmp_flag(flag)
            t0 = time.perf_counter()
            _, fmin_ref, code_ref = R.opt_lbfgs_forces(np.zeros(M), w0, yT, YTilde, th, LBFGS_DEFAULTS)
            rec["reference"].append({"fast_openmp": flag, "code": int(code_ref), "fmin": float(fmin_ref),
                                     "seconds": time.perf_counter() - t0})
        R.set_fast_openmp_flag(1)
        rec["reference_code"] = rec["reference"][0]["code"]
        rec["reference_fmin"] = rec["reference"][0]["fmin"]
        rec["fmin_rel_diff"] = max(abs(dev.fmin - r["fmin"]) / abs(r["fmin"]) for r in rec["reference"])
        codes = {r["code"] for r in rec["reference"]}
        rec["status_in_reference_spread"] = int(dev.lbfgs_code) in codes
        rec["rounding_floor_ending"] = bool((codes | {int(dev.lbfgs_code)}) <= set(floor_codes) and rec["fmin_rel_diff"] <= 1e-12)
        out["per_theta"].append(rec)
    out["same_status_everywhere"] = all(r["reference_code"] == r["device_code"] for r in out["per_theta"])
    out["status_in_reference_spread_everywhere"] = all(r["status_in_reference_spread"] for r in out["per_theta"])
    out["same_minimum_everywhere"] = all(r["fmin_rel_diff"] <= 1e-12 for r in out["per_theta"])
    out["every_difference_at_the_rounding_floor"] = all(r["rounding_floor_ending"] for r in out["per_theta"]
                                                        if r["reference_code"] != r["device_code"])
    return out


def deer_trace(d_nm, t_ns):
    """DEER / PELDOR form factor of one spin pair at distance d (Fresnel form, the kernel of the reference's rotamer
    example: examples/DEER/rotamer-refinement/POTRA/bioen_rotamer.py:113-143):
        F(d, t) = [C(x) cos(pi x^2 / 6) + S(x) sin(pi x^2 / 6)] / x,   x = sqrt(6 D t / (pi d^3)),  D = 2 pi 52.04e-3 nm^3/ns,
    C, S the Fresnel integrals (Abramowitz & Stegun convention, scipy.special.fresnel); F(d, 0) = 1.
    d_nm: (N,), t_ns: (M,)  ->  (M, N)"""
    from scipy.special import fresnel
    D = 2.0 * np.pi * 52.04e-3
    t = np.asarray(t_ns, dtype=np.float64)[:, None]
    d = np.asarray(d_nm, dtype=np.float64)[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        x = np.sqrt(6.0 * D * t / (np.pi * d ** 3))
        sf, cf = fresnel(x)
        F = (cf * np.cos(np.pi / 6.0 * x * x) + sf * np.sin(np.pi / 6.0 * x * x)) / x
    F[t[:, 0] == 0.0, :] = 1.0
    return F


def deer_inputs(N, seed, sigma=0.01):
    """BASELINE configs[3] as SURVEY 8(d) defines it: N rotamer distances (seeded mixture in 2-6 nm), the 205-point
    time axis and the measured signal of exp-370-292-signal-deer.dat (tests/golden/deer_exp_370_292.npz: data taken
    from the reference tree in the build container), traces by deer_trace, sigma = 0.01 (run_bioen.py:306).
    -> (Ft = (F - 1) / sigma  [M x N, the modulation-depth-independent matrix], YTilde = signal / sigma, off = 1 / sigma)"""
    z = np.load(os.path.join(ROOT, "tests", "golden", "deer_exp_370_292.npz"))
    t_ns = z["t_us"] * 1000.0
    rng = np.random.default_rng(seed)
    d = np.where(rng.random(N) < 0.6, rng.normal(3.2, 0.35, N), rng.normal(4.8, 0.5, N)).clip(2.0, 6.0)   # nm
    M = t_ns.size
    Ft = np.empty((M, N))

    def chunk(c0):                                 # (the ufuncs release the GIL: column chunks in threads)
        blk = deer_trace(d[c0:c0 + 16384], t_ns)
        blk -= 1.0
        blk /= sigma
        Ft[:, c0:c0 + 16384] = blk
    from concurrent.futures import ThreadPoolExecutor
    try:
        nthreads = max(1, min(32, len(os.sched_getaffinity(0))))
    except AttributeError:
        nthreads = 4
    with ThreadPoolExecutor(nthreads) as pool:
        list(pool.map(chunk, range(0, N, 16384)))
    return Ft, z["signal"] / sigma, np.full(M, 1.0 / sigma)


def deer_record(bioen_amd, seed, with_cpu=True):
    """BASELINE configs[3] on one GPU: DEER refinement with a modulation-depth nuisance parameter, N = 5e5 rotamers
    x M = 205 time points of the measured trace exp-370-292 (y~(m) = 1/sigma + m (F - 1)/sigma, F the Fresnel
    form of the reference's rotamer example).  The m-independent matrix (F - 1)/sigma is resident; per theta the
    reference alternates `iterations` = 10 times between a BioEn optimisation and a 1-D least-squares refit of m that
    REBUILDS y~ on the host (procedure.py:62-83, observables.py:110-171), the modulation depth starting at 0.15
    (run_bioen.py:296) and carried across thetas (procedure.py:82-83); here a refit is 2 m doubles back from the device
    and a closed form, the new m enters through the affine row model."""
    from bioen_amd import nuisance
    N = 500000
    t0 = time.perf_counter()
    Ft, YT, off = deer_inputs(N, seed)
    t_inputs = time.perf_counter() - t0
    M = Ft.shape[0]
    m0 = 0.15
    G = np.zeros(N)
    thetas, iterations = [100.0, 10.0, 1.0], 10
    with bioen_amd.Context(Ft, YT) as ctx:
        nuisance.series(ctx, thetas[:1], G, G, LBFGS_DEFAULTS, YT, row_offset=off, scale0=m0, iterations=1)      # warm-up, builds the strip copies
        ctx.kernel_stats_enable(True)
        ctx.kernel_stats_reset()
        ctx.synchronize()
        t0 = time.perf_counter()
        res = nuisance.series(ctx, thetas, G, G, LBFGS_DEFAULTS, YT, row_offset=off, scale0=m0, iterations=iterations)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        stats = ctx.kernel_stats()
        ctx.kernel_stats_enable(False)
        try:
            ceil_gbs = ctx.read_probe(reps=10)[0]
        except Exception:
            ceil_gbs = None
    its = int(sum(sum(x["iterations"] for x in r["trace"]) for r in res))
    evs = int(sum(sum(x["evaluations"] for x in r["trace"]) for r in res))
    # roofline of the matrix kernels at M = 205 (one problem per launch: the series runs theta by theta): algorithmic bytes
    # of a launch = the matrix once + one N-vector and one M-vector in, one out, as at the headline
    kern = {}
    for name, kname in (("forward", "k_strip_fwd"), ("adjoint", "k_strip_adj")):
        st = stats[name]
        launches = max(st["launches"], 1)
        avg_ms = st["total_ms"] / launches
        avg_k = st["problem_passes"] / launches
        alg = float(M) * N * 8 + avg_k * (8.0 * N + 8.0 * M)
        kern[name] = {"kernel": kname, "launches": st["launches"], "avg_ms": avg_ms, "avg_batch_width": avg_k,
                      "algorithmic_bytes": alg, "achieved_GBs": alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0}
    dom = max(kern, key=lambda k: kern[k]["avg_ms"])
    roofline = {"bound": "hbm", "kernel": kern[dom]["kernel"], "achieved": kern[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kern[dom]["achieved_GBs"] / HBM_PEAK_GBS, "traffic": None, "kernels": kern,
                "note": "M = 205 is stored as 208 rows per strip (the 16-row operand block): 1.5 % more bytes are read than "
                        "the algorithmic count credits",
                "read_ceiling_GBs": ceil_gbs}
    cpu = None
    if with_cpu:
        try:
            cpu = deer_cpu_baseline(Ft, YT, off, m0)
        except Exception as e:
            cpu = {"value": None, "unit": "iter*N*M/s", "cores": 0, "kind": "error", "sample": repr(e)}
    return {"roofline": roofline, "cpu_baseline": cpu, "workload": "DEER rotamer refinement, N=%d x M=%d (trace exp-370-292, Fresnel kernel), thetas %s, %d optimise/refit "
                        "iterations each (cold-started weights, modulation depth carried over), yaml-default liblbfgs, "
                        "modulation depth refitted on the resident matrix" % (N, M, thetas, iterations),
            "seconds": dt, "iterations": its, "evaluations": evs, "value": its * float(N) * M / dt, "unit": "iter*N*M/s",
            "refits": len(thetas) * iterations, "moddepth_start": m0, "input_synthesis_s": t_inputs,
            "moddepth_fit": [r["scales"][0] for r in res], "fmin": [r["fmin"] for r in res],
            "chi2": [r["chi2"] for r in res]}


def deer_cpu_baseline(Ft, YT, off, m0, cols=131072, theta=10.0, iterations=2, cap=60):
    """configs[3]'s protocol the way the reference runs it (observables.py:110-171, 205-210; procedure.py:62-83), on a
    BOUNDED sample: the first `cols` rotamers, one theta, `iterations` optimise/refit rounds -- every round REBUILDS
    yTilde(m) = 1/sigma + m (F - 1)/sigma on the host, runs the reference's own _opt_lbfgs_logw (oracle/_ref; capped at
    `cap` iterations) on it with its transposed cache built per call (c_bioen.pyx:471-473), and refits m on chi^2 from a
    host GEMV.  value = L-BFGS iterations * cols * M / wall, rebuilds and refits included."""
    from oracle import ref_binding as R
    from oracle import cpus
    from bioen_amd import nuisance
    if not R.available():
        return None
    cores = cpus.usable_cpus()
    R.set_fast_openmp_flag(1)
    R.omp_set_num_threads(cores)
    F = np.ascontiguousarray(Ft[:, :cols])
    M, n = F.shape
    G = np.zeros(n)
    params = dict(LBFGS_DEFAULTS, max_iterations=cap)
    R.logw_f(G, G, F, YT, theta)                      # thread pool up
    m, its, t_rebuild, t_opt, uncounted = m0, 0, 0.0, 0.0, []
    t0 = time.perf_counter()
    for _ in range(iterations):
        t1 = time.perf_counter()
        explicit = off[:, None] + m * F               # the host rebuild of yTilde(m)
        t_rebuild += time.perf_counter() - t1
        t1 = time.perf_counter()
        g, fmin, code = R.opt_lbfgs_logw(G, G, explicit, YT, theta, params)
        t_opt += time.perf_counter() - t1
        # (the reference's driver counts its iterations in a C global it does not return: -997 <=> the cap was reached, else
        # the restatement's count on the same inputs, untimed -- as cpu_baseline above)
        uncounted.append(None if code == -997 else explicit)
        its += cap if code == -997 else 0
        w = R.get_weights(g)[0]
        m = float(nuisance.refit_scales(F.dot(w), YT, off, [np.arange(M)])[0])
    dt = time.perf_counter() - t0
    for explicit in uncounted:
        if explicit is not None:
            from oracle import oracle_binding as O
            its += O.opt_lbfgs_logw(G, G, explicit, YT, theta, params)[3]
    return {"value": its * float(n) * M / dt, "unit": "iter*N*M/s", "cores": cores, "kind": "reference",
            "sample": "first %d of the rotamers x M=%d, theta=%g, %d optimise/refit rounds of at most %d L-BFGS iterations each "
                      "(%d in all), host rebuild of yTilde(m) every round: %.2f s (rebuilds %.2f s, _opt_lbfgs_logw %.2f s)"
                      % (n, M, theta, iterations, cap, its, dt, t_rebuild, t_opt),
            "seconds": dt, "iterations": its, "moddepth_fit": m}


ALA5_LBFGS = dict(linesearch=2, max_iterations=20000, delta=1e-6, epsilon=1e-5, ftol=1e-4, gtol=0.9, wolfe=0.9, past=10,
                  max_linesearch=100)          # examples/ala5_optimize/lbfgs_2.yaml:34-49


def ala5_record(bioen_amd, seed, with_cpu=True):
    """The only timing inside the reference tree: the ala5 notebook's forces series -- N = 50001
    structures x M = 28 observables, 80 thetas logspace(5, -1) (examples/ala5_optimize/thetas2.dat), liblbfgs with the
    settings of lbfgs_2.yaml, every theta WARM-started from the previous optimum (ala5-bioen.ipynb, run_theta_series:
    `forces_init = forces_opt`), 29.1 s in all on the notebook's workstation.  Same shape, settings and protocol on
    synthetic data: the device (80 dependent single-problem runs, launch-bound), the reference's own C path on this
    box's cores, and the cold-started series as ONE lock-step batch on the device."""
    N, M = 50001, 28
    thetas = np.logspace(5, -1, 80)
    YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M, seed)
    w0 = np.full(N, 1.0 / N)
    out = {"workload": "forces theta series, N=%d x M=%d, 80 thetas logspace(5,-1), liblbfgs settings of lbfgs_2.yaml, "
                       "warm starts (ala5-bioen.ipynb protocol), synthetic data" % (N, M),
           "reference_notebook_s": 29.1}
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=seed) as ctx:
        def warm_series():
            f, fmins, its, codes = np.zeros(M), [], 0, []
            for th in thetas:
                f, _, info = ctx.opt_lbfgs_forces(f, w0, th, ALA5_LBFGS, want_weights=True)
                fmins.append(info.fmin)
                codes.append(info.lbfgs_code)
                its += info.iterations
            return fmins, its, codes
        warm_series()                                   # builds the strip copy, warms the allocator
        ctx.synchronize()
        t0 = time.perf_counter()
        fmins, its, codes = warm_series()
        ctx.synchronize()
        out["gpu_warm_series_s"] = time.perf_counter() - t0
        out["gpu_iterations"] = its
        out["gpu_codes_not_0_1_2"] = sorted(set(c for c in codes if c not in (0, 1, 2)))
        t0 = time.perf_counter()
        _, _, infos = ctx.opt_lbfgs_forces_batch(thetas, np.zeros(M), w0, ALA5_LBFGS, max_batch=8)
        ctx.synchronize()
        out["gpu_cold_batched_s"] = time.perf_counter() - t0
        out["gpu_cold_iterations"] = int(sum(i.iterations for i in infos))
        yT = np.ascontiguousarray(ctx.read_ytilde()) if with_cpu else None
    if with_cpu:
        from oracle import ref_binding as R
        from oracle import cpus
        if R.available():
            cores = cpus.usable_cpus()
            R.set_fast_openmp_flag(1)
            R.omp_set_num_threads(cores)
            R.forces_f(np.zeros(M), w0, yT, YTilde, 1.0)
            f, rel, rcodes, t0 = np.zeros(M), [], [], time.perf_counter()
            for th, fm in zip(thetas, fmins):
                f, fmin, code = R.opt_lbfgs_forces(f, w0, yT, YTilde, th, ALA5_LBFGS)     # transposed cache per call, as c_bioen.pyx
                rel.append(abs(fm - fmin) / abs(fmin))
                rcodes.append(code)
            out["cpu_codes_not_0_1_2"] = sorted(set(c for c in rcodes if c not in (0, 1, 2)))
            out["cpu_warm_series_s"] = time.perf_counter() - t0
            out["cpu_cores"] = cores
            out["speedup"] = out["cpu_warm_series_s"] / out["gpu_warm_series_s"]
            out["fmin_rel_diff_max"] = max(rel)
    return out


def api_record(bioen_amd, yTilde, YTilde, thetas, label, with_unheld):
    """"theta-sweep wall-clock (upload/generation excluded AND included, both stated)".  The headline `value` has
    the matrix generated in HBM; this is what a caller of the kept Python API pays who holds a HOST numpy matrix
    (bioen/analyze/procedure.py:62-77's shape): wall time INCLUDING the upload and the two strip-copy builds, and the
    number of uploads, for
      series        one `log_weights.find_optimum_series` call (all thetas one lock-step batch),
      analyze_loop  `find_optimum` per theta inside ``with optimize.resident(yTilde):`` -- one upload for the loop,
      unheld_loop   the same loop without the hold: one upload per call (r04: three per call)  [smaller size only],
    each result the 5-tuple of the reference's API (weights, averages, optimum, fmin_initial, fmin_final)."""
    from bioen_amd import optimize
    from bioen_amd.optimize.ext import c_bioen
    c_bioen.clear_cache()
    M, N = yTilde.shape
    G = np.zeros((N, 1))
    YT = np.asarray(YTilde, dtype=np.float64).reshape(1, -1)
    cfg = optimize.minimize.Parameters("lbfgs")
    cfg["verbose"] = False
    rec = {"workload": "%s: host numpy yTilde %d x %d (%.2f GB), %d thetas, bioen_amd.optimize.log_weights API, yaml-default lbfgs"
                       % (label, M, N, yTilde.nbytes / 1e9, len(thetas)), "N": N, "M": M, "host_matrix_bytes": int(yTilde.nbytes)}

    def timed(fn):
        u0, t0 = c_bioen.uploads, time.perf_counter()
        out = fn()
        return out, time.perf_counter() - t0, c_bioen.uploads - u0

    out, dt, up = timed(lambda: optimize.log_weights.find_optimum_series(G, G, yTilde, yTilde, YT, thetas, cfg))
    its = sum(i.iterations for i in c_bioen.last_opt_info)
    rec["series"] = {"wall_s": dt, "uploads": up, "iterations": int(its), "value_incl_upload": its * float(N) * M / dt,
                     "fmin": [float(o[4]) for o in out],
                     "per_theta": [{"theta": float(t), "iterations": int(i.iterations), "evaluations": int(i.evaluations),
                                    "code": int(i.lbfgs_code)} for t, i in zip(thetas, c_bioen.last_opt_info)]}
    c_bioen.clear_cache()

    def loop(held):
        per, its_ = [], 0
        def body():
            nonlocal its_
            for th in thetas:
                t0 = time.perf_counter()
                optimize.log_weights.find_optimum(G, G, yTilde, yTilde, YT, float(th), cfg)
                per.append(time.perf_counter() - t0)
                its_ += c_bioen.last_opt_info.iterations
        if held:
            with optimize.resident(yTilde):
                body()
        else:
            body()
        return per, its_
    (per, its), dt, up = timed(lambda: loop(True))
    rec["analyze_loop"] = {"wall_s": dt, "uploads": up, "iterations": int(its), "per_theta_s": per,
                           "value_incl_upload": its * float(N) * M / dt}
    c_bioen.clear_cache()
    if with_unheld:
        (per, its), dt, up = timed(lambda: loop(False))
        rec["unheld_loop"] = {"wall_s": dt, "uploads": up, "iterations": int(its)}
        c_bioen.clear_cache()
    return rec


class stdout_to_stderr(object):
    """librccl prints a version banner on stdout when a communicator is created; keep this
    process' stdout for the ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def choose_transport(ctx, comm, sweep, nshard, transport, count, xinfo, HipError, quiet=stdout_to_stderr):
    """What the cross-rank traffic of a multi-rank run goes through -> (gather description, rccl in use).  Every decision is
    taken on values all ranks have all-gathered, so every rank takes the same branch.

    theta-dealing (nshard False): the results travel once, at the end: RCCL if every rank can initialise it, else TCP.
    Structure sharding: the stage exchanges of the rounds.  auto = the peer-to-peer mailboxes where they attach (every
    rank maps every peer's mailbox and passes the self-test) -- no collective library on the path at all; RCCL where they
    do not; host-staged as the last resort.  compare = p2p and RCCL both measured, the faster one kept.  Fills xinfo with
    what was measured ({p2p,rccl,host}_us: the slowest rank's microseconds per exchange, None where a probe failed)."""
    def probe():          # slowest rank's view; inf where the transport does not work
        try:
            t = ctx.exchange_probe(count=count, reps=40)
        except HipError as e:
            xinfo.setdefault("probe_errors", []).append(str(e))
            t = float("inf")
        return max(comm.allgather_object(t))

    def start_rccl():     # -> (rccl, gather); the outcome agreed between the ranks
        try:
            with quiet():
                ok = sweep.init_rccl(ctx, comm)
            how = "rccl-allgather"
        except HipError as e:   # report, keep the control-plane path
            ok, how = False, "tcp-allgather (RCCL unavailable: %s)" % e
        if not all(comm.allgather_object(bool(ok))):
            if ok:                                # this rank has a communicator the others cannot use
                ctx.comm_destroy()
                how = "tcp-allgather (RCCL init failed on some rank)"
            elif not how.startswith("tcp"):
                how = "tcp-allgather (RCCL init failed on this rank)"
            ok = False
        return ok, how

    gather, rccl = "none", False
    if not nshard:
        rccl, gather = start_rccl()
        return gather, rccl
    p2p = False
    rccl_tried = False
    if transport == "compare":
        rccl, gather = start_rccl()
        rccl_tried = True
        if rccl:
            xinfo["rccl_us"] = probe()
    if transport in ("auto", "p2p", "compare"):
        p2p = sweep.init_p2p(ctx, comm)          # agreed between the ranks; self-tested
        xinfo["p2p_attached"] = p2p
        if p2p:
            xinfo["p2p_us"] = probe()
            if transport == "compare" and xinfo.get("rccl_us", float("inf")) < xinfo["p2p_us"]:
                ctx.p2p_detach()                 # RCCL is the faster one on this node
                p2p = False
    if not p2p and transport in ("auto", "rccl") and not rccl_tried:
        rccl, gather = start_rccl()
        rccl_tried = True
        if rccl:
            xinfo["rccl_us"] = probe()
    stage_rccl = rccl and not p2p                # what the rounds' stage exchanges go through
    if not p2p and not rccl:
        ctx.set_exchange(comm)   # host-staged all-gathers: correct but slow (ranks sharing one GPU)
        xinfo["host_us"] = probe()
    # r05: whatever carries the stage exchanges, EVERY multi-rank run owns an RCCL communicator over all its ranks for the
    # final all-gather of (fmin, chi^2, S, iterations, status, weights) per theta (north_star: "an RCCL gather over xGMI of
    # the final (S, chi^2, weights) per theta") -- initialised here, outside the timed region; the mailboxes keep the
    # in-loop exchanges (the library prefers them: api.hip: exchange).  If RCCL cannot be had the run goes on, and says so.
    if not rccl_tried and transport != "host":     # (host forced: a communicator would take the stage exchanges over)
        rccl, gather = start_rccl()
    elif not rccl_tried:
        gather = "tcp-allgather (--transport host: RCCL not initialised)"
    xinfo["rccl_gather"] = bool(rccl)
    if not rccl:
        xinfo["rccl_error"] = gather
    if not stage_rccl:
        gather = "stage exchanges (every rank holds the results) + %s" % ("one rccl-allgather of the results" if rccl else gather)
    xinfo["transport"] = ctx.exchange_transport()
    xinfo["exchange_us"] = xinfo.get({"p2p": "p2p_us", "rccl": "rccl_us", "host": "host_us"}[xinfo["transport"]])
    for k in list(xinfo):
        if isinstance(xinfo[k], float) and not np.isfinite(xinfo[k]):
            xinfo[k] = None
    return gather, rccl


class WallBudget(object):
    """Wall budget of bench.py's side records (r06): `seconds` from its creation; `reserve` = what the records that are
    never dropped still need.  take(name, estimate) -> may this droppable record still run; timed(name) times one;
    report() -> what ran for how long and what was left out."""

    def __init__(self, seconds, reserve=0.0, clock=time.perf_counter):
        self.clock, self.t0, self.seconds, self.reserve = clock, clock(), float(seconds), float(reserve)
        self.spent, self.dropped = {}, []

    def left(self):
        return self.seconds - (self.clock() - self.t0)

    def take(self, name, estimate):
        if self.left() - self.reserve < estimate:
            self.dropped.append(name)
            return False
        return True

    def skipped(self, estimate):
        return {"skipped": "budget", "estimate_s": estimate, "budget_left_s": round(self.left(), 1), "reserved_s": round(self.reserve, 1)}

    def timed(self, name):
        budget = self

        class _T(object):
            def __enter__(self_):
                self_.t = budget.clock()

            def __exit__(self_, *exc):
                budget.spent[name] = round(budget.spent.get(name, 0.0) + budget.clock() - self_.t, 2)
                return False
        return _T()

    def report(self):
        return {"budget_s": self.seconds, "used_s": round(self.clock() - self.t0, 1), "seconds": dict(self.spent),
                "skipped_for_budget": list(self.dropped)}


def visible_devices(python=None, root=ROOT, timeout=180.0):
    """Number of GPUs HIP shows -- asked in a CHILD process, so that the caller (the rank launcher below) never initialises
    the GPU itself.  -> (count, None) or (None, reason)"""
    import subprocess
    code = "import sys; sys.path.insert(0, %r); import bioen_amd; print('BIOEN_DEVICES', bioen_amd.device_count())" % root
    try:
        p = subprocess.run([python or os.path.realpath(sys.executable), "-c", code], capture_output=True, text=True, timeout=timeout)
    except Exception as e:
        return None, repr(e)
    for ln in p.stdout.splitlines():
        if ln.startswith("BIOEN_DEVICES "):
            return int(ln.split()[1]), None
    return None, "device probe: exit %d: %s" % (p.returncode, (p.stderr or p.stdout)[-300:].strip())


def launch_ranks(nranks, child_argv, timeout, out=None, err=None, port=None, poll=0.05):
    """`python3 bench.py --gpus N` without a launcher around it (r06): this process becomes the launcher.  It starts N fresh
    child processes of `child_argv` -- one rank each, the environment torchrun would give them (RANK, LOCAL_RANK, WORLD_SIZE,
    LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT, a run id) -- relays rank 0's stdout (the ONE JSON line; the
    other ranks' stdout goes to stderr), and returns the exit status: 0 if every rank left with 0, else the FIRST non-zero
    status seen (the remaining ranks are terminated at once: nothing waits on a dead peer), 124 after `timeout` seconds
    (all ranks killed).  No HIP call, no torch, no bioen_amd import happens in this process: the children own the GPUs.
    Replaces the serial theta loop of the reference (bioen/analyze/procedure.py:62-63) on the launcher side."""
    import signal
    import socket
    import subprocess
    import threading
    out = out or sys.stdout
    err = err or sys.stderr
    if port is None:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    run_id = "bench_%d_%d" % (os.getpid(), int(time.time()))
    children, pumps = [], []

    def pump(src, dst):
        for ln in iter(src.readline, ""):
            dst.write(ln)
            dst.flush()
        src.close()

    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHELASTIC_RUN_ID=run_id, BIOEN_BENCH_LAUNCHED="1")
        p = subprocess.Popen(child_argv, env=env, stdout=subprocess.PIPE, text=True, bufsize=1, start_new_session=True)
        children.append(p)
        t = threading.Thread(target=pump, args=(p.stdout, out if r == 0 else err), daemon=True)
        t.start()
        pumps.append(t)

    def stop_all(sig):
        for p in children:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)          # the rank and whatever it started (its rocprofv3 passes)
                except (ProcessLookupError, PermissionError):
                    pass

    deadline = time.time() + timeout
    status = 0
    try:
        while True:
            codes = [p.poll() for p in children]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                status = bad[0] if bad[0] > 0 else 128 - bad[0]      # (a rank killed by signal s: 128 + s, as a shell reports it)
                which = codes.index(bad[0])
                print("bench.py: rank %d left with status %d; stopping the other ranks" % (which, status), file=err)
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                status = 124
                print("bench.py: the ranks did not finish within %.0f s; killing them" % timeout, file=err)
                break
            time.sleep(poll)
    finally:
        if any(p.poll() is None for p in children):
            stop_all(signal.SIGTERM)
            t_end = time.time() + 5.0
            while time.time() < t_end and any(p.poll() is None for p in children):
                time.sleep(poll)
            stop_all(signal.SIGKILL)
        for p in children:
            try:
                p.wait(timeout=10.0)
            except Exception:
                pass
        for t in pumps:
            t.join(timeout=5.0)
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--structures", type=int, default=1000000, help="N (default: BASELINE configs[2])")
    ap.add_argument("--method", choices=("logw", "forces"), default="logw",
                    help="logw: BASELINE configs[2] (default); forces: configs[4] (forces method, M = 512 unless given)")
    ap.add_argument("--observables", type=int, default=None, help="M (default 1024; 512 with --method forces)")
    ap.add_argument("--thetas", type=int, default=8, help="points of the theta series")
    ap.add_argument("--max-batch", type=int, default=8, help="thetas sharing one matrix pass (1 = unbatched)")
    ap.add_argument("--shard", choices=("auto", "structures", "thetas"), default="auto",
                    help="multi-GPU decomposition: split the N structures (columns) of every pass, or deal "
                         "thetas; auto = structures when the measured all-gather latency makes it the faster one")
    ap.add_argument("--transport", choices=("auto", "p2p", "rccl", "host", "compare"), default="auto",
                    help="stage exchanges of a structure-sharded run: peer-to-peer mailboxes (hipIpc over xGMI, one kernel "
                         "per all-gather), RCCL all-gathers, or host-staged; auto = the mailboxes where they attach and pass "
                         "their self-test, else RCCL, else host-staged; compare = p2p and RCCL both measured, the faster taken")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-forces", action="store_true", help="skip the forces-method record (configs[4])")
    ap.add_argument("--no-matched", action="store_true", help="skip the matched CPU/GPU sweep at configs[1] size")
    ap.add_argument("--no-deer", action="store_true", help="skip the DEER nuisance-refit record (configs[3])")
    ap.add_argument("--no-ala5", action="store_true", help="skip the ala5-shaped forces series (BASELINE.md's only reference timing)")
    ap.add_argument("--cpu-cols", type=int, default=524288, help="columns of the matrix the CPU baseline runs on")
    ap.add_argument("--cpu-iters", type=int, default=120, help="L-BFGS iterations the CPU baseline is capped at")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the live rocprofv3 --pmc passes for roofline.traffic (falls back to profiles/traffic.json)")
    ap.add_argument("--no-storage-experiment", action="store_true",
                    help="skip the reduced-byte storage side record (fp32 + bf16 split / fp32 copies of the matrix)")
    ap.add_argument("--no-one-copy", action="store_true", help="skip the side record of the sweep with ONE strip copy resident")
    ap.add_argument("--no-cpu-mid", action="store_true",
                    help="skip the full-size converged parity run of the reference at theta = 31.6 (~1.5-2 minutes of CPU)")
    ap.add_argument("--no-api", action="store_true",
                    help="skip the api_end_to_end record (the kept Python API on a host numpy matrix, upload included)")
    ap.add_argument("--no-cpu-fullsize", action="store_true",
                    help="skip the reference run on the FULL headline matrix (two cheapest thetas, ~1 minute)")
    ap.add_argument("--share-devices", action="store_true",
                    help="allow more ranks than visible GPUs (rank r on device r %% devices): a TEST mode for one-GPU boxes -- "
                         "without it a run with fewer devices than --gpus exits with status 5 and one line saying so")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="seconds after which the self-launched ranks of --gpus N > 1 are killed (exit status 124)")
    ap.add_argument("--budget", type=float, default=260.0,
                    help="wall budget in seconds for the SIDE records after the timed region; they are dropped lowest priority "
                         "first with \"skipped\": \"budget\" -- roofline and cpu_baseline are never dropped")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # r06: no launcher around us -- become it.  Nothing below this branch runs in this process: no HIP call, no
        # bioen_amd import; the N children re-enter main() with the rank environment set.
        ndev, why = visible_devices()
        if ndev is None:
            print("bench.py: cannot count the GPUs (%s)" % why, file=sys.stderr)
            sys.exit(5)
        if ndev < args.gpus and not args.share_devices:
            print("bench.py: --gpus %d but only %d device(s) visible to HIP; not running (--share-devices puts several ranks "
                  "on one device: a test mode, labelled as such in the line)" % (args.gpus, ndev), file=sys.stderr)
            sys.exit(5)
        sys.exit(launch_ranks(args.gpus, [os.path.realpath(sys.executable), os.path.abspath(__file__)] + sys.argv[1:],
                              args.launch_timeout))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        # the label and the ranks must agree: a line saying n_gpus: N is N ranks, never fewer
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to run under a wrong label" % (args.gpus, world), file=sys.stderr)
        sys.exit(5)

    import bioen_amd
    from bioen_amd import sweep

    forces_mode = args.method == "forces"
    N = args.structures
    M = args.observables if args.observables else (512 if forces_mode else 1024)
    thetas = np.logspace(3, -0.5, args.thetas)
    YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M)

    comm = sweep.SocketComm() if world > 1 else sweep.SingleComm()
    ndev = bioen_amd.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no MI355X visible to HIP -- this benchmark has no CPU path")
    if ndev < world and not args.share_devices:
        if rank == 0:
            print("bench.py: %d ranks but only %d device(s) visible to HIP; not running (--share-devices: test mode)"
                  % (world, ndev), file=sys.stderr)
        sys.exit(5)
    xinfo = {}     # what the stage exchanges of the sharded context go through, and what each transport measured

    def build(nshard):
        """context + communicator for one of the two decompositions"""
        ctx = bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED,
                                          device=local_rank % ndev, rank=rank if nshard else 0,
                                          world=world if nshard else 1)
        gather, rccl = "none", False
        xinfo.clear()
        if world > 1:
            gather, rccl = choose_transport(ctx, comm, sweep, nshard, args.transport, M * min(8, len(thetas)), xinfo,
                                            bioen_amd.BioenHipError)
        return ctx, gather, rccl

    decision = None
    nshard = world > 1 and args.shard in ("auto", "structures")
    ctx, gather, rccl = build(nshard)
    if world > 1 and args.shard == "auto":
        # Splitting the structures pays when the per-pass saving beats the 3 small all-gathers a
        # round then needs (ybar + softmax totals, gradient dots, Gram update) plus launch
        # overhead; decided with margin:  t_pass * (1 - 1/world)  vs  5 * t_exchange + 0.15 ms.
        t_ex = xinfo.get("exchange_us")            # measured by build(): the chosen transport, slowest rank
        probe_error = "; ".join(xinfo.get("probe_errors", [])) or None
        if t_ex is None:                           # an exchange that does not work anywhere: deal thetas everywhere
            t_ex = float("inf")
        t_pass_us = 2.0 * M * float(N) * 8 / 6.4e12 * 1e6
        gain_us = t_pass_us * (1.0 - 1.0 / world)
        cost_us = (3.0 * t_ex + 100.0) if forces_mode else (4.0 * t_ex + 150.0)   # 2 exchanges per round, with margin
        decision = {"exchange_us": t_ex if np.isfinite(t_ex) else None, "transport": xinfo.get("transport"),
                    "pass_saving_us": gain_us,
                    "exchange_cost_us": cost_us if np.isfinite(cost_us) else None,
                    "chosen": "structures" if gain_us > cost_us else "thetas", "probe_error": probe_error}
        if gain_us <= cost_us:
            ctx.close()
            nshard = False
            xsaved = dict(xinfo)
            ctx, gather, rccl = build(False)
            xinfo.update(xsaved, transport=None)   # keep what was measured; nothing is exchanged when thetas are dealt

    G = np.zeros(N)          # w0 = 1/N  =>  G = 0 ; GInit = G
    g0 = np.zeros(N)
    w0 = np.full(N, 1.0 / N)
    f0 = np.zeros(M)

    def step():
        if forces_mode:
            if nshard:     # configs[4]'s decomposition: every rank a column block, all thetas batched everywhere
                return sweep.sweep_forces_sharded(ctx, thetas, w0, f0, LBFGS_DEFAULTS, max_batch=args.max_batch)
            return sweep.sweep_forces(ctx, thetas, w0, f0, LBFGS_DEFAULTS, comm=comm, rccl=rccl, max_batch=args.max_batch)
        if nshard:
            # every rank holds a column block of yTilde and takes part in every theta of the batch
            return sweep.sweep_log_weights_sharded(ctx, thetas, G, g0, LBFGS_DEFAULTS, max_batch=args.max_batch)
        return sweep.sweep_log_weights(ctx, thetas, G, g0, LBFGS_DEFAULTS, comm=comm, rccl=rccl,
                                       max_batch=args.max_batch)

    results = None
    warm_done = 0
    if nshard:
        # one untimed sharded sweep first: if the collectives fail on any rank, every rank falls back to
        # dealing thetas (no communication inside the loop) instead of losing the measurement
        try:
            results, ok, why = step(), True, None
        except bioen_amd.BioenHipError as e:
            ok, why = False, str(e)
        if all(comm.allgather_object(ok)):
            warm_done = 1
        else:
            ctx.close()
            nshard = False
            xsaved = dict(xinfo)
            ctx, gather, rccl = build(False)
            xinfo.update(xsaved, transport=None)
            decision = dict(decision or {}, chosen="thetas", fallback="structure-sharded sweep failed: %s" % why)
    for _ in range(max(args.warmup - warm_done, 0)):
        results = step()

    # (structure-sharded: did a rank fall back to the one-copy form alone?  then the bits are not those of other GPU counts)
    forms = sweep.check_common_form(ctx, comm) if nshard else None
    ctx.kernel_stats_enable(True)
    ctx.kernel_stats_reset()
    comm.barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    try:
        for _ in range(args.steps):
            results = step()
        ctx.synchronize()
    except bioen_amd.BioenHipError as e:
        # a transport that failed inside the timed region (a rank gone, a wait past BIOEN_HIP_WAIT_TIMEOUT): every rank
        # gets here through the bounded waits; say so in ONE line and leave with a non-zero status -- nothing is retried
        # in a process whose GPU work has failed
        if rank == 0:
            print(json.dumps({"metric": "L-BFGS iterations/sec x (N structures * M observables), theta sweep", "value": None,
                              "unit": "iter*N*M/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "error": str(e), "config": {"exchange_transport": xinfo.get("transport"),
                                                          "sharding_fallback": True}}))
            sys.stdout.flush()
        sys.exit(4)
    comm.barrier()
    dt = comm.max(time.perf_counter() - t0)
    stats = ctx.kernel_stats()
    ctx.kernel_stats_enable(False)
    ctx_layout = ctx.layout()
    resident_forms, resident_bytes = ctx.footprint()        # (r06: one strip copy of a matrix above 1 GiB -- 8.2 GB at the headline, not 16.4)

    # r05: the final (S, chi^2, weights) per theta once more through ONE RCCL all-gather over all ranks -- in a
    # structure-sharded run the cross-rank consistency check (every rank must hold the same bytes), after the timed region;
    # theta-dealing runs have gathered through RCCL inside it (sweep.theta_sweep) and are checked the same way.  A failure
    # is reported in the line, never fatal.
    final_gather = None
    if world > 1:
        # only the gather itself may fail on a rank; the two control-plane collectives behind it run on EVERY rank whatever
        # happened, so that no rank waits in them for one that left through the except branch
        try:
            final_gather = sweep.gather_results(ctx, results, comm, rccl=bool(rccl))
        except bioen_amd.BioenHipError as e:
            final_gather = {"error": str(e), "via": "rccl" if rccl else "tcp", "seconds": 0.0, "consistent": False}
        final_gather["seconds"] = comm.max(final_gather["seconds"])
        states = comm.allgather_object([bool(final_gather["consistent"]), final_gather.get("error")])
        final_gather["consistent"] = bool(all(s_[0] for s_ in states))
        errors = [s_[1] for s_ in states if s_[1]]
        if errors:
            final_gather["error"] = errors[0]
            final_gather["ranks_failed"] = len(errors)

    iters_per_sweep = sum(r["iterations"] for r in results)
    evals_per_sweep = sum(r["evaluations"] for r in results)
    total_iters = iters_per_sweep * args.steps
    value = total_iters * float(N) * M / dt

    if rank == 0:
        # ---- roofline of the dominant (slower) matrix-streaming kernel, rank 0's launches ----
        # algorithmic bytes of ONE launch serving K thetas: the matrix once, plus per theta one
        # N-vector and one M-vector in, one out (matrix bytes are shared by the batch)
        n_rank = ctx.n_local if nshard else N            # columns streamed by one launch on rank 0
        # M > 1024 (log-weights): one launch per row panel of <= 1024 rows, each streaming its share of the matrix
        panels = 1 if (M <= 1024 or forces_mode or os.environ.get("BIOEN_HIP_PANELS") == "0") else (M + 1023) // 1024
        mat_bytes = float(M) * n_rank * 8 / panels
        kern = {}
        for name in ("forward", "adjoint"):
            s = stats[name]
            launches = max(s["launches"], 1)
            avg_ms = s["total_ms"] / launches
            avg_k = s["problem_passes"] / launches
            alg = mat_bytes + avg_k * (8.0 * n_rank + 8.0 * M)
            strip = (M <= 1024 or panels > 1) and not os.environ.get("BIOEN_HIP_FWD_STREAM") == "1"     # kernels_strip.hip
            if forces_mode:      # timer slots of launch_forces_xy ("adjoint") / _bt ("forward")
                kname = "%s<K, nt, %s>" % ("k_strip" if M <= 512 else "k_strip2", "true" if name == "adjoint" else "false") if M <= 1024 else \
                        ("k_adj + k_fwd_partial" if name == "adjoint" else "k_fwd_partial")
            else:
                one_adj = "k_strip" if M <= 512 else "k_strip2"          # the one-copy adjoint: the forces kernels' ADJ form
                kname = ("k_strip_fwd" if strip else "k_fwd_partial") if name == "forward" else \
                        ((one_adj if ctx_layout["one_copy"] else "k_strip_adj") if strip else "k_adj")
            kern[name] = {"kernel": kname,
                          "launches": s["launches"], "avg_ms": avg_ms, "avg_batch_width": avg_k,
                          "algorithmic_bytes": alg,
                          "achieved_GBs": alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0}
        dom = max(kern, key=lambda k: kern[k]["avg_ms"])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.isfile(tpath):
            try:
                with open(tpath) as fp:
                    tj = json.load(fp)
                key = "%s_N%d_M%d" % ("k_strip" if forces_mode else kern[dom]["kernel"], n_rank, M)
                # only a PMC pass of THESE kernel sources counts; a stale file yields null
                traffic = tj.get(key) if tj.get("_source_sha") == kernel_source_sha() else None
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": kern[dom]["kernel"], "achieved": kern[dom]["achieved_GBs"],
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kern[dom]["achieved_GBs"] / HBM_PEAK_GBS,
                    "traffic": traffic, "kernels": kern,
                    "traffic_source": "profiles/traffic.json (builder's rocprofv3 --pmc passes of these kernel sources)"
                                      if traffic is not None else None}

        try:      # what this box's memory system gives a plain read-only stream over the same copy of the matrix
            ceil_gbs, ceil_bytes = ctx.read_probe(reps=10)
            roofline["read_ceiling"] = {"GB/s": ceil_gbs, "bytes_per_pass": ceil_bytes, "frac_of_peak": ceil_gbs / HBM_PEAK_GBS,
                                        "kernel": "k_read_probe: wide nontemporal loads and an add, no LDS, no matrix cores",
                                        "achieved_over_ceiling": roofline["achieved"] / ceil_gbs}
        except Exception as e:
            roofline["read_ceiling"] = {"error": repr(e)}

        # ---- side records, under a wall budget (r06) -------------------------------------------------------------
        # Everything below is reported BESIDE the headline; the driver's clock runs on.  `--budget` seconds (default 260)
        # from here: the records run in priority order and one that no longer fits -- by its estimate, with the time the
        # never-dropped ones still need held back -- is left out with "skipped": "budget".  Never dropped: cpu_baseline
        # (the bounded sample) and roofline.traffic (the live counter passes).
        budget = WallBudget(args.budget, reserve=0.0)
        single = world == 1
        want_pmc = single and not args.no_pmc and M <= 1024
        budget.reserve = (20.0 if want_pmc else 0.0) + (25.0 if single and not args.no_cpu_baseline else 0.0)
        stats_per_sweep = {k: {kk: vv / max(args.steps, 1) for kk, vv in stats[k].items()} for k in stats}

        cpu = None
        if single and not args.no_cpu_baseline:
            with budget.timed("cpu_baseline"):
                try:
                    if forces_mode:
                        cpu = cpu_baseline_forces(ctx, M, N, YTilde, 10.0, min(args.cpu_cols, 262144), 40)
                    else:
                        cpu = cpu_baseline(ctx, M, N, YTilde, 10.0, args.cpu_cols, args.cpu_iters)
                except Exception as e:   # the baseline is a reported extra; never lose the GPU line over it
                    cpu = {"value": None, "unit": "iter*N*M/s", "cores": 0, "kind": "error", "sample": repr(e)}
            budget.reserve -= 25.0

        forces = None                    # priority 1: configs[4]
        if single and not args.no_forces and not forces_mode:
            if budget.take("forces", 60.0):
                with budget.timed("forces"):
                    try:
                        forces = forces_record(bioen_amd, thetas, SEED, args.max_batch, with_cpu=not args.no_cpu_baseline,
                                                with_split=not args.no_storage_experiment and budget.left() - budget.reserve > 120.0)
                        tj_f = None
                        if os.path.isfile(tpath):
                            with open(tpath) as fp:
                                tj_f = json.load(fp)
                        if tj_f and tj_f.get("_source_sha") == kernel_source_sha():
                            forces["roofline"]["traffic"] = tj_f.get("k_strip_N1000000_M512")
                    except Exception as e:
                        forces = {"error": repr(e)}
            else:
                forces = budget.skipped(60.0)

        deer = None                      # priority 2: configs[3]
        if single and not args.no_deer and not args.no_forces and not forces_mode:
            if budget.take("deer", 15.0):
                with budget.timed("deer"):
                    try:
                        deer = deer_record(bioen_amd, SEED, with_cpu=not args.no_cpu_baseline)
                    except Exception as e:
                        deer = {"error": repr(e)}
            else:
                deer = budget.skipped(15.0)

        if cpu is not None and single and not args.no_matched and not args.no_cpu_baseline and not forces_mode:
            if budget.take("matched_sweep", 45.0):          # priority 3: configs[1], numpy's stream to the letter, against the reference
                with budget.timed("matched_sweep"):
                    try:
                        cpu["matched_sweep"] = cpu_matched(bioen_amd, thetas, SEED)
                        if cpu["matched_sweep"] and "single_thread" in cpu["matched_sweep"]:
                            cpu["single_thread"] = cpu["matched_sweep"].pop("single_thread")
                    except Exception as e:
                        cpu["matched_sweep"] = {"error": repr(e)}
            else:
                cpu["matched_sweep"] = budget.skipped(45.0)

        mids = []
        if cpu is not None and single and not forces_mode and not args.no_cpu_fullsize and N * float(M) >= 5e8:
            if budget.take("cpu_full_size", 35.0):          # priority 4: the same CONFIG on the CPU, not a sample
                with budget.timed("cpu_full_size"):
                    try:     # the two cheapest thetas of the series; then, while the budget lasts, a mid-series one
                        cheap = sorted(results, key=lambda r: r["evaluations"])[:2]
                        # the mid-series thetas nearest 31.6 first (the cheap ones above excluded)
                        mids = [] if args.no_cpu_mid else sorted((r["theta"] for r in results if r not in cheap and 2.0 < r["theta"] < 60.0),
                                                               key=lambda t: abs(np.log(t / 31.6)))
                        cpu["full_size"] = cpu_fullsize(ctx, M, N, YTilde, [r["theta"] for r in cheap])      # (a mid-series theta: last record below)
                    except Exception as e:
                        cpu["full_size"] = {"error": repr(e)}
            else:
                cpu["full_size"] = budget.skipped(35.0)

        api = None                       # priority 5: the kept Python API on a host matrix, upload included
        if single and not forces_mode and not args.no_api:
            if budget.take("api_end_to_end", 45.0):
                with budget.timed("api_end_to_end"):
                    try:
                        # the headline SIZE on the host as a caller of the Python API holds it -- r06: SURVEY 8(d)'s numpy
                        # stream to the letter (PCG64(12345), row-wise normals: 1.02e9 draws on one host thread, ~15 s), so that
                        # iteration counts on the survey's own inputs stand beside the device generator's of the headline
                        t_gen = time.perf_counter()
                        host_matrix, host_YT = survey_inputs(M, N)
                        t_gen = time.perf_counter() - t_gen
                        api = {"headline": api_record(bioen_amd, host_matrix, host_YT, thetas,
                                                      "BASELINE configs[2] on SURVEY 8(d)'s numpy stream (PCG64 seed %d, row-wise normals)" % SEED, False)}
                        api["headline"]["host_generation_s"] = t_gen
                        del host_matrix
                        y1, Y1 = survey_inputs(256, 100000)
                        api["configs1"] = api_record(bioen_amd, y1, Y1, thetas, "BASELINE configs[1] (SURVEY 8(d)'s numpy stream)", True)
                        del y1
                        api["kernel_only_sweep_s"] = dt / max(args.steps, 1)
                    except Exception as e:
                        api = dict(api or {}, error=repr(e))
            else:
                api = budget.skipped(45.0)

        one_copy = None                  # priority 6
        if single and not forces_mode and M <= 1024 and not args.no_one_copy:
            if budget.take("one_copy", 5.0):
                with budget.timed("one_copy"):
                    try:
                        one_copy = one_copy_record(bioen_amd, M, N, YTrue, sig_sim, sig_exp, YTilde, thetas, args.max_batch, results,
                                                   dt / max(args.steps, 1), stats_per_sweep, bool(ctx_layout["one_copy"]))
                    except Exception as e:
                        one_copy = {"error": repr(e)}
            else:
                one_copy = budget.skipped(5.0)

        storage = None                   # priority 7: the reduced-byte storage experiment
        if single and not forces_mode and M <= 1024 and not args.no_storage_experiment:
            if budget.take("storage_experiment", 12.0):
                with budget.timed("storage_experiment"):
                    try:
                        sweep_s = dt / max(args.steps, 1)
                        storage = storage_record(ctx, thetas, G, g0, args.max_batch, results, stats_per_sweep)
                        storage["f64"]["sweep_s"] = sweep_s
                        storage["f64"]["ms_per_round"] = 1e3 * sweep_s / max(storage["f64"]["rounds"], 1)
                    except Exception as e:
                        storage = {"error": repr(e)}
            else:
                storage = budget.skipped(12.0)

        ala5 = None                      # priority 8
        if single and not args.no_ala5 and not args.no_forces and not forces_mode:
            if budget.take("ala5", 5.0):
                with budget.timed("ala5"):
                    try:
                        ala5 = ala5_record(bioen_amd, SEED, with_cpu=not args.no_cpu_baseline)
                    except Exception as e:
                        ala5 = {"error": repr(e)}
            else:
                ala5 = budget.skipped(5.0)

        # priority 9, with whatever the budget has left: a theta of the series' expensive half on the FULL matrix, the device's
        # converged optimum put before the reference (cpu_fullsize: mid_theta) -- ~35 s of the reference's yaml-default run,
        # 30 s of the device's converged one, two short reference runs at the optimum
        if (cpu is not None and isinstance(cpu.get("full_size"), dict) and "per_theta" in cpu["full_size"] and mids
                and not args.no_cpu_mid):
            if budget.take("cpu_mid_theta", 75.0):
                with budget.timed("cpu_mid_theta"):
                    try:
                        more = cpu_fullsize(ctx, M, N, YTilde, [], mid_thetas=mids,
                                            mid_budget_s=max(0.0, budget.left() - budget.reserve - 35.0),      # (its own device runs: ~35 s)
                                            ms_per_eval=cpu["full_size"].get("cpu_ms_per_evaluation"))
                        cpu["full_size"]["mid_theta"] = (more or {}).get("mid_theta")
                    except Exception as e:
                        cpu["full_size"]["mid_theta"] = {"error": repr(e)}
            else:
                cpu["full_size"]["mid_theta"] = budget.skipped(75.0)

        if want_pmc:
            # roofline.traffic from counters read on THIS box, in this run (the committed profiles/traffic.json stays the
            # fallback): child processes, after every context of this one is closed.  Never dropped.
            budget.reserve = 0.0
            ctx.close()
            with budget.timed("roofline_traffic"):
                base = roofline["kernel"].split("<")[0]
                t_live, src = live_traffic(args.method, M, N, base)
                if t_live is not None:
                    roofline["traffic"], roofline["traffic_source"] = t_live, src
                else:
                    roofline["traffic_live_error"] = src
                if forces and "roofline" in forces and budget.left() > 25.0:
                    t_live, src = live_traffic("forces", 512, 1000000, "k_strip")
                    if t_live is not None:
                        forces["roofline"]["traffic"], forces["roofline"]["traffic_source"] = t_live, src

        line = {
            "metric": "L-BFGS iterations/sec x (N structures * M observables), %s theta sweep"
                      % ("forces-method" if forces_mode else "log-weights"),
            "value": value,
            "unit": "iter*N*M/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / max(args.steps, 1),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s theta sweep, N=%d structures x M=%d observables, %d thetas "
                                   "logspace(3,-0.5), cold starts, liblbfgs yaml defaults; yTilde = SURVEY 8(d)'s recipe "
                                   "(row-wise normals around YTrue, sig_sim = 0.5 YTrue, sig_exp = 0.1 YTrue) drawn by a "
                                   "device counter-based normal stream in HBM (k_generate: SplitMix64 + Box-Muller, seed %d), "
                                   "not numpy's PCG64 stream -- iteration counts are specific to this generator "
                                   "(cpu_baseline.matched_sweep runs numpy's stream to the letter)"
                                   % ("forces-method" if forces_mode else "log-weights", N, M, len(thetas), SEED),
                       "method": args.method,
                       "N": N, "M": M, "thetas": [float(t) for t in thetas], "lbfgs": LBFGS_DEFAULTS,
                       "sharding": ("structures (columns) split over %d rank(s), all thetas batched on every rank" % world)
                       if nshard else ("theta round-robin over %d rank(s)" % world), "gather": gather,
                       "max_batch": args.max_batch, "shard_decision": decision,
                       "rccl_ranks": world if rccl else (0 if world > 1 else None),
                       "rccl_error": xinfo.get("rccl_error") if (world > 1 and not rccl) else None,
                       "rccl_gather_us": (1e6 * final_gather["seconds"] if final_gather and final_gather.get("via") == "rccl"
                                          and "seconds" in final_gather else None),
                       "final_gather": final_gather,
                       "exchange_transport": xinfo.get("transport") if nshard else None,
                       "exchange_us": xinfo.get("exchange_us") if xinfo else (decision or {}).get("exchange_us"),
                       "exchange_us_by_transport": {k[:-3]: xinfo[k] for k in ("p2p_us", "rccl_us", "host_us") if k in xinfo},
                       "decomposition": ("structures" if nshard else "thetas") if world > 1 else "single GPU",
                       "strip_copy_forms": forms, "strip_layout": ctx_layout,
                       "resident": sorted(resident_forms), "resident_bytes": resident_bytes,
                       "devices_visible": ndev, "ranks_share_devices": bool(world > ndev),
                       "launched_by": "bench.py itself (subprocess ranks)" if os.environ.get("BIOEN_BENCH_LAUNCHED") == "1"
                                      else ("an outer launcher (torchrun environment)" if world > 1 else "single process"),
                       "sharding_fallback": bool(world > 1 and ((decision or {}).get("fallback") or
                                                                not (nshard and xinfo.get("transport") in ("p2p", "rccl"))))},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "forces": forces,
            "storage_experiment": storage,
            "one_copy": one_copy,
            "api_end_to_end": api,
            "deer": deer,
            "ala5": ala5,
            "side_records": budget.report(),
            "sweep_wall_s": dt / max(args.steps, 1),
            "iterations_per_sweep": iters_per_sweep,
            "evaluations_per_sweep": evals_per_sweep,
            "per_theta": [{"theta": r["theta"], "rank": r["rank"], "iterations": r["iterations"],
                           "evaluations": r["evaluations"], "code": r["code"], "fmin": r["fmin"],
                           "chi2": r["chi2"], "S": r["S"], "seconds": r["seconds"]} for r in results],
        }
        print(json.dumps(line))
        sys.stdout.flush()

    # --shard structures was asked for explicitly: a run that ended on anything but RCCL-backed structure sharding has
    # not measured what was asked -- the line above says so (sharding_fallback), the exit status too
    failed = world > 1 and args.shard == "structures" and not (nshard and xinfo.get("transport") in ("p2p", "rccl"))
    ctx.close()
    comm.close()
    if failed:
        print("bench.py: --shard structures requested, but the run fell back (transport: %s, decomposition: %s)"
              % (xinfo.get("transport"), "structures" if nshard else "thetas"), file=sys.stderr)
    # (a rank whose RCCL initialisation was abandoned at its time bound leaves through os._exit: _lib.leave_process)
    from bioen_amd import _lib as _l
    _l.leave_process(3 if failed else 0)


if __name__ == "__main__":
    main()
