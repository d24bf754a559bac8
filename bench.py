#!/usr/bin/env python3
"""bench.py -- BioEn log-weights theta-sweep on MI355X (BASELINE.json metric).

One "step" = one complete pass of the hot path over the workload: the 8-point
theta series (np.logspace(3, -0.5, 8), every theta cold-started, liblbfgs yaml
defaults) of the log-weights optimizer on a synthetic N = 1e6 x M = 1024 ensemble
(BASELINE.json configs[2]; the matrix is generated in HBM, so inputs are resident
when the timed region starts).  With --gpus N the thetas are sharded over the N
ranks (one process per GPU, launched by torch.distributed.run; this script itself
is torch-free) and the per-theta results are all-gathered over RCCL/xGMI.

value = (L-BFGS iterations of the whole job) * N * M / wall-clock, max over ranks.

    python bench.py                      # 1 GPU, 1 warm-up + 1 timed sweep
    python bench.py --gpus 1 --steps 2 --warmup 1
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


def synthetic_targets(M, seed=SEED):
    """Per-observable vectors of the SURVEY 8(d) recipe (after forces.py:19-68)."""
    rng = np.random.default_rng(seed)
    YTrue = rng.uniform(1, 10, M)
    sig_exp = 0.1 * YTrue
    sig_sim = 0.5 * YTrue
    YTilde = rng.normal(YTrue, sig_exp) / sig_exp
    return YTrue, sig_sim, sig_exp, YTilde


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--structures", type=int, default=1000000, help="N (default: BASELINE configs[2])")
    ap.add_argument("--observables", type=int, default=1024, help="M")
    ap.add_argument("--thetas", type=int, default=8, help="points of the theta series")
    ap.add_argument("--max-batch", type=int, default=8, help="thetas sharing one matrix pass (1 = unbatched)")
    ap.add_argument("--shard", choices=("auto", "structures", "thetas"), default="auto",
                    help="multi-GPU decomposition: split the N structures (columns) of every pass, or deal "
                         "thetas; auto = structures when the measured all-gather latency makes it the faster one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cols", type=int, default=524288, help="columns of the matrix the CPU baseline runs on")
    ap.add_argument("--cpu-iters", type=int, default=120, help="L-BFGS iterations the CPU baseline is capped at")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    import bioen_amd
    from bioen_amd import sweep

    N, M = args.structures, args.observables
    thetas = np.logspace(3, -0.5, args.thetas)
    YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M)

    comm = sweep.SocketComm() if world > 1 else sweep.SingleComm()
    ndev = bioen_amd.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no MI355X visible to HIP -- this benchmark has no CPU path")
    def build(nshard):
        """context + communicator for one of the two decompositions"""
        ctx = bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED,
                                          device=local_rank % ndev, rank=rank if nshard else 0,
                                          world=world if nshard else 1)
        gather, rccl = "none", False
        if world > 1:
            try:
                with stdout_to_stderr():
                    rccl = sweep.init_rccl(ctx, comm)
                gather = "rccl-allgather"
            except bioen_amd.BioenHipError as e:   # report, keep the control-plane path
                gather = "tcp-allgather (RCCL unavailable: %s)" % e
                rccl = False
            if not all(comm.allgather_object(rccl)):
                rccl = False
                gather = "tcp-allgather (RCCL init failed on some rank)"
                ctx.comm_destroy()
            if nshard and not rccl:
                ctx.set_exchange(comm)   # host-staged all-gathers: correct but slow (ranks sharing one GPU)
        return ctx, gather, rccl

    decision = None
    nshard = world > 1 and args.shard in ("auto", "structures")
    ctx, gather, rccl = build(nshard)
    if world > 1 and args.shard == "auto":
        # Splitting the structures pays when the per-pass saving beats the 3 small all-gathers a
        # round then needs (ybar + softmax totals, gradient dots, Gram update) plus launch
        # overhead; decided with margin:  t_pass * (1 - 1/world)  vs  5 * t_exchange + 0.15 ms.
        try:
            t_mine, probe_error = ctx.exchange_probe(count=M * min(8, len(thetas)), reps=40), None
        except bioen_amd.BioenHipError as e:      # an exchange that does not work anywhere: deal thetas everywhere
            t_mine, probe_error = float("inf"), str(e)
        t_ex = max(comm.allgather_object(t_mine))
        t_pass_us = 2.0 * M * float(N) * 8 / 6.4e12 * 1e6
        gain_us = t_pass_us * (1.0 - 1.0 / world)
        cost_us = 5.0 * t_ex + 150.0
        decision = {"exchange_us": t_ex if np.isfinite(t_ex) else None, "pass_saving_us": gain_us,
                    "exchange_cost_us": cost_us if np.isfinite(cost_us) else None,
                    "chosen": "structures" if gain_us > cost_us else "thetas", "probe_error": probe_error}
        if gain_us <= cost_us:
            ctx.close()
            nshard = False
            ctx, gather, rccl = build(False)

    G = np.zeros(N)          # w0 = 1/N  =>  G = 0 ; GInit = G
    g0 = np.zeros(N)

    def step():
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
            ctx, gather, rccl = build(False)
            decision = dict(decision or {}, chosen="thetas", fallback="structure-sharded sweep failed: %s" % why)
    for _ in range(max(args.warmup - warm_done, 0)):
        results = step()

    ctx.kernel_stats_enable(True)
    ctx.kernel_stats_reset()
    comm.barrier()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        results = step()
    ctx.synchronize()
    comm.barrier()
    dt = comm.max(time.perf_counter() - t0)
    stats = ctx.kernel_stats()
    ctx.kernel_stats_enable(False)

    iters_per_sweep = sum(r["iterations"] for r in results)
    evals_per_sweep = sum(r["evaluations"] for r in results)
    total_iters = iters_per_sweep * args.steps
    value = total_iters * float(N) * M / dt

    if rank == 0:
        # ---- roofline of the dominant (slower) matrix-streaming kernel, rank 0's launches ----
        # algorithmic bytes of ONE launch serving K thetas: the matrix once, plus per theta one
        # N-vector and one M-vector in, one out (matrix bytes are shared by the batch)
        n_rank = ctx.n_local if nshard else N            # columns streamed by one launch on rank 0
        mat_bytes = float(M) * n_rank * 8
        kern = {}
        for name in ("forward", "adjoint"):
            s = stats[name]
            launches = max(s["launches"], 1)
            avg_ms = s["total_ms"] / launches
            avg_k = s["problem_passes"] / launches
            alg = mat_bytes + avg_k * (8.0 * n_rank + 8.0 * M)
            kern[name] = {"kernel": "k_fwd_partial" if name == "forward" else "k_adj",
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
                key = "%s_N%d_M%d" % (kern[dom]["kernel"], n_rank, M)
                traffic = tj.get(key)
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": kern[dom]["kernel"], "achieved": kern[dom]["achieved_GBs"],
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kern[dom]["achieved_GBs"] / HBM_PEAK_GBS,
                    "traffic": traffic, "kernels": kern}

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(ctx, M, N, YTilde, 10.0, args.cpu_cols, args.cpu_iters)
            except Exception as e:   # the baseline is a reported extra; never lose the GPU line over it
                cpu = {"value": None, "unit": "iter*N*M/s", "cores": 0, "kind": "error", "sample": repr(e)}

        line = {
            "metric": "L-BFGS iterations/sec x (N structures * M observables), log-weights theta sweep",
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
            "config": {"workload": "log-weights theta sweep, N=%d structures x M=%d observables, %d thetas "
                                   "logspace(3,-0.5), cold starts, liblbfgs yaml defaults" % (N, M, len(thetas)),
                       "N": N, "M": M, "thetas": [float(t) for t in thetas], "lbfgs": LBFGS_DEFAULTS,
                       "sharding": ("structures (columns) split over %d rank(s), all thetas batched on every rank" % world)
                       if nshard else ("theta round-robin over %d rank(s)" % world), "gather": gather,
                       "max_batch": args.max_batch, "shard_decision": decision},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "sweep_wall_s": dt / max(args.steps, 1),
            "iterations_per_sweep": iters_per_sweep,
            "evaluations_per_sweep": evals_per_sweep,
            "per_theta": [{"theta": r["theta"], "rank": r["rank"], "iterations": r["iterations"],
                           "evaluations": r["evaluations"], "code": r["code"], "fmin": r["fmin"],
                           "chi2": r["chi2"], "S": r["S"], "seconds": r["seconds"]} for r in results],
        }
        print(json.dumps(line))
        sys.stdout.flush()

    ctx.close()
    comm.close()


if __name__ == "__main__":
    main()
