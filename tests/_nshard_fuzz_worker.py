"""Worker of tests/test_hip_nshard.py::test_randomised_series_on_sharded_contexts: one rank of a structure-sharded context
running random theta series (tools/fuzz_batch.py's recipe at sharded sizes): every problem of a batch must return the bits
of its single run ON THIS CONTEXT, every rank the same bits, nothing non-finite -- with the ranks' host threads pausing at
random and one rank's delivery threads dawdling (BIOEN_HIP_JITTER_US / _DELIVERY_US set by the test)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bioen_amd                      # noqa: E402
from bioen_amd import sweep            # noqa: E402
from conftest import LBFGS_DEFAULTS    # noqa: E402


def sig(x, w, info):
    bits = lambda v: np.float64(v).tobytes()
    return (x.tobytes(), w.tobytes(), bits(info.fmin), info.iterations, info.evaluations, info.lbfgs_code)


def main():
    out_path, nseeds = sys.argv[1], int(sys.argv[2])
    comm = sweep.SocketComm()
    bad, digest = [], hashlib.sha256()
    first = int(os.environ.get("NSHARD_FUZZ_FIRST", "0"))
    for seed in range(first, first + nseeds):
        rng = np.random.default_rng(7000 + seed)
        M = int(rng.choice([int(v) for v in os.environ.get("NSHARD_FUZZ_M", "16,64,96,205,512,600").split(",")]))
        N = int(rng.choice([int(v) for v in os.environ.get("NSHARD_FUZZ_N", "1000,2049,5000").split(",")]))
        YTrue = rng.uniform(1, 10, M)
        y = rng.normal(YTrue[:, None], 0.5 * YTrue[:, None], (M, N)) / (0.1 * YTrue[:, None])
        YT = rng.normal(YTrue, 0.1 * YTrue) / (0.1 * YTrue)
        nt = int(rng.integers(1, 12))
        thetas = 10.0 ** rng.uniform(-1.0, 3.0, nt)
        max_batch = int(rng.integers(1, 9))
        G = np.log(rng.dirichlet(np.ones(N) * 2.0))
        w0 = rng.dirichlet(np.ones(N) * 2.0)
        shared = rng.random() < 0.5
        g0 = G if shared else np.stack([G + 0.05 * k * rng.standard_normal(N) for k in range(nt)])
        f0 = np.zeros(M) if shared else np.stack([1e-4 * k * rng.standard_normal(M) for k in range(nt)])
        params = dict(LBFGS_DEFAULTS, linesearch=int(rng.choice([0, 1, 2, 3])), max_iterations=int(rng.integers(2, 30)))
        tag = "seed %d: M=%d N=%d thetas=%d batch=%d ls=%d it<=%d" % (seed, M, N, nt, max_batch, params["linesearch"], params["max_iterations"])
        ctx = bioen_amd.Context(y, YT, device=0, rank=comm.rank, world=comm.world)
        try:
            assert sweep.init_p2p(ctx, comm), "the peer-to-peer exchange did not attach"
            res, w, infos = ctx.opt_lbfgs_logw_batch(thetas, g0, G, params, max_batch=max_batch)
            with_forces = M <= 1024            # (the forces method on a sharded context: strip passes only)
            if with_forces:
                fres, fw, finfos = ctx.opt_lbfgs_forces_batch(thetas, f0, w0, params, max_batch=max_batch)
            else:
                fres, fw, finfos = res[:, :M], w, infos
            for k in range(nt):
                one = ctx.opt_lbfgs_logw(g0 if shared else g0[k], G, thetas[k], params)
                if sig(res[k], w[k], infos[k]) != sig(*one):
                    bad.append("%s: log-weights problem %d differs from its single run" % (tag, k))
                fone = ctx.opt_lbfgs_forces(f0 if shared else f0[k], w0, thetas[k], params) if with_forces else (fres[k], fw[k], finfos[k])
                if sig(fres[k], fw[k], finfos[k]) != sig(*fone):
                    bad.append("%s: forces problem %d differs from its single run" % (tag, k))
                if not (np.isfinite(res[k]).all() and np.isfinite(fres[k]).all()):
                    bad.append("%s: problem %d returned non-finite numbers" % (tag, k))
                for part in sig(res[k], w[k], infos[k]) + sig(fres[k], fw[k], finfos[k]):
                    digest.update(part if isinstance(part, bytes) else repr(part).encode())
        except Exception as e:
            bad.append(tag + " EXCEPTION " + repr(e)[:300])
            break                                  # a failed exchange leaves the ranks out of step: stop here
        finally:
            ctx.close()
    digests = comm.allgather_object(digest.hexdigest()) if not bad else [digest.hexdigest()]
    with open(out_path % comm.rank, "w") as fp:
        fp.write(repr({"bad": bad, "digests": digests}))
    comm.close()


if __name__ == "__main__":
    main()
