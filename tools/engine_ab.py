"""Development aid: A/B of the log-weights engine variants in ONE process on ONE context (boxes differ by several
percent, and by 20 % at mid sizes): the environment switches are read at every call, so the variants are interleaved
sweep by sweep.  SIZES="M:N,..."  REPS=3.  Prints the best and the median sweep time per variant."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
from bioen_amd import sweep
from bench import synthetic_targets, LBFGS_DEFAULTS, SEED

VARIANTS = {
    "device": {},
    "device-nospec": {"BIOEN_HIP_SPECULATE": "0"},
    "device-eager": {"BIOEN_HIP_SHADOW_RATE": "0", "BIOEN_HIP_SHADOWS": "8"},
    "device-noqueue": {"BIOEN_HIP_QUEUE": "0"},
    "device-queue2": {"BIOEN_HIP_QUEUE": "2"},
    # the policy of sharded contexts (two slots kept back, both steps of the slowest thetas from the first search on), on one GPU
    "device-reserve": {"BIOEN_HIP_SHADOWS": "2", "BIOEN_HIP_SHADOW_RATE": "0", "BIOEN_HIP_SHADOW_MINEV": "0", "BIOEN_HIP_DEV_RESERVE": "2"},
    "device-shadows0": {"BIOEN_HIP_SHADOWS": "2", "BIOEN_HIP_SHADOW_RATE": "0", "BIOEN_HIP_SHADOW_MINEV": "0", "BIOEN_HIP_DEV_RESERVE": "0"},
    # ... in the form sharded contexts run it (shadows sweep their own Gram products), and with the stage exchanges of the
    # sharded round executed through the peer-to-peer transport with ONE rank: the exchange kernels are launched (two per
    # round) and find nobody to wait for -- the launch share of an exchange, on one GPU
    "sharded-form": {"BIOEN_HIP_SHADOWS": "2", "BIOEN_HIP_SHADOW_RATE": "0", "BIOEN_HIP_SHADOW_MINEV": "0", "BIOEN_HIP_DEV_RESERVE": "2",
                     "BIOEN_HIP_SHADOW_GRAM": "1"},
    "sharded-form+p2p1": {"BIOEN_HIP_SHADOWS": "2", "BIOEN_HIP_SHADOW_RATE": "0", "BIOEN_HIP_SHADOW_MINEV": "0", "BIOEN_HIP_DEV_RESERVE": "2",
                          "BIOEN_HIP_SHADOW_GRAM": "1", "_P2P": "1"},
    # the device-resident engine FORCED (above 4 GB of matrix traffic per round the default is the host-driven one)
    "dev1": {"BIOEN_HIP_DEVICE_LS": "1"},
    "dev1-reserve": {"BIOEN_HIP_DEVICE_LS": "1", "BIOEN_HIP_SHADOWS": "2", "BIOEN_HIP_SHADOW_RATE": "0", "BIOEN_HIP_SHADOW_MINEV": "0",
                     "BIOEN_HIP_DEV_RESERVE": "2"},
    "dev1-sharded-form": {"BIOEN_HIP_DEVICE_LS": "1", "BIOEN_HIP_SHADOWS": "2", "BIOEN_HIP_SHADOW_RATE": "0", "BIOEN_HIP_SHADOW_MINEV": "0",
                          "BIOEN_HIP_DEV_RESERVE": "2", "BIOEN_HIP_SHADOW_GRAM": "1"},
    # ... with more shadow slots (the two slowest thetas shadowed once slots come free / from the start)
    "dev1-sharded-form-sh4": {"BIOEN_HIP_DEVICE_LS": "1", "BIOEN_HIP_SHADOWS": "4", "BIOEN_HIP_SHADOW_RATE": "0", "BIOEN_HIP_SHADOW_MINEV": "0",
                              "BIOEN_HIP_DEV_RESERVE": "2", "BIOEN_HIP_SHADOW_GRAM": "1"},
    "dev1-sharded-form-sh4r4": {"BIOEN_HIP_DEVICE_LS": "1", "BIOEN_HIP_SHADOWS": "4", "BIOEN_HIP_SHADOW_RATE": "0", "BIOEN_HIP_SHADOW_MINEV": "0",
                                "BIOEN_HIP_DEV_RESERVE": "4", "BIOEN_HIP_SHADOW_GRAM": "1"},
    "dev1-sharded-form-sh6": {"BIOEN_HIP_DEVICE_LS": "1", "BIOEN_HIP_SHADOWS": "6", "BIOEN_HIP_SHADOW_RATE": "0", "BIOEN_HIP_SHADOW_MINEV": "0",
                              "BIOEN_HIP_DEV_RESERVE": "2", "BIOEN_HIP_SHADOW_GRAM": "1"},
    "host": {"BIOEN_HIP_DEVICE_LS": "0"},
    "host-nospec": {"BIOEN_HIP_DEVICE_LS": "0", "BIOEN_HIP_SPECULATE": "0"},
}
KEYS = sorted({k for v in VARIANTS.values() for k in v if not k.startswith('_')})
sizes = [tuple(int(v) for v in s.split(":")) for s in os.environ.get("SIZES", "256:100000").split(",")]
reps = int(os.environ.get("REPS", "3"))
only = os.environ.get("VARIANTS")
names = [n for n in VARIANTS if not only or n in only.split(",")]
thetas = np.logspace(3, -0.5, 8)
for (M, N) in sizes:
    YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M)
    world = int(os.environ.get("WORLD", "1"))          # WORLD=8: rank 0 of an 8-rank decomposition alone on this GPU, its
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED, rank=0, world=world) as ctx:
        if world > 1:                                  # exchanges mirrored (Context.set_mirror_exchange): a rank's share per round
            ctx.set_mirror_exchange(True)
        G = np.zeros(N)
        times = {n: [] for n in names}
        rounds = {}
        for rep in range(reps + 1):
            for n in names:
                for k in KEYS:
                    os.environ.pop(k, None)
                os.environ.update({k: v for k, v in VARIANTS[n].items() if not k.startswith("_")})
                p2p = VARIANTS[n].get("_P2P") == "1"
                if p2p:
                    ctx.p2p_export()
                    ctx.p2p_attach(None)
                    ctx.set_force_exchange(True)
                ctx.kernel_stats_enable(rep == reps)
                ctx.kernel_stats_reset()
                ctx.synchronize()
                t0 = time.perf_counter()
                res = sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS)
                ctx.synchronize()
                dt = time.perf_counter() - t0
                if p2p:
                    ctx.set_force_exchange(False)
                    ctx.p2p_detach()
                if rep == reps:
                    rounds[n] = ctx.kernel_stats()["forward"]["launches"]
                elif rep > 0 or reps == 1:
                    times[n].append(dt)
        for n in names:
            t = sorted(times[n])
            print(("WORLD=%d rank share  " % world if world > 1 else "") + "M=%d N=%d %-16s best %.4f s  median %.4f s  rounds %d -> %.1f us/round" % (
                M, N, n, t[0], t[len(t) // 2], rounds[n], 1e6 * t[0] / max(rounds[n], 1)))
        if world > 1:
            # the sweep's fixed cost (start-up, eight result deliveries with their gathers) weighs on the few hundred rounds of
            # the mirrored problem: the same series cut after one iteration measures it; the rest is the rounds' own time
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update({k: v for k, v in VARIANTS[names[0]].items() if not k.startswith("_")})
            short = dict(LBFGS_DEFAULTS, max_iterations=1)
            t1 = []
            for rep in range(4):
                ctx.kernel_stats_enable(True)
                ctx.kernel_stats_reset()
                ctx.synchronize()
                t0 = time.perf_counter()
                sweep.sweep_log_weights(ctx, thetas, G, G, short)
                ctx.synchronize()
                t1.append(time.perf_counter() - t0)
                r1 = ctx.kernel_stats()["forward"]["launches"]
            t = sorted(times[names[0]])
            print("WORLD=%d rank share  %-16s the series cut after one iteration: %.4f s for %d rounds -> net of that fixed cost "
                  "%.1f us/round" % (world, names[0], min(t1[1:]), r1, 1e6 * (t[0] - min(t1[1:])) / max(rounds[names[0]] - r1, 1)))
        sys.stdout.flush()
