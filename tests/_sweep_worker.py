"""Worker of tests/test_sweep_distributed.py: one rank of a 2-process theta sweep on the CPU.
The solver injected into bioen_amd.sweep.theta_sweep is the ORACLE (this is a test of the
sharding / gather logic, which has no GPU dependence); the product passes its HIP solver."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from bioen_amd import sweep          # noqa: E402
from oracle import oracle_binding as O   # noqa: E402
from conftest import load_golden, LBFGS_DEFAULTS   # noqa: E402


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    if mode == "gloo":
        import torch.distributed as dist
        dist.init_process_group("gloo")
        comm = sweep.TorchComm()
    else:
        comm = sweep.SocketComm()
    d = load_golden("synth_logw_M37xN500.npz")
    thetas = [50.0, 0.5, 5.0, 500.0, 1.0]      # 5 thetas on 2 ranks: uneven shards, padded gather

    def solve(theta):
        g, fmin, code, it, ev = O.opt_lbfgs_logw(d["GInit"], d["G"], d["yTilde"], d["YTilde"], theta, LBFGS_DEFAULTS)
        f, _, w = O.logw_fdf(g, d["G"], d["yTilde"], d["YTilde"], theta)
        chi2, _ = O.chi_squared(w, d["yTilde"], d["YTilde"])
        info = types.SimpleNamespace(fmin=fmin, chi2=chi2, kl=(fmin - chi2) / theta, iterations=it, evaluations=ev,
                                     lbfgs_code=code, seconds=0.0)
        return w, info

    res = sweep.theta_sweep(None, thetas, solve, comm=comm, rccl=False, n=d["G"].size)
    tmax = comm.max(float(comm.rank + 1))
    # the final all-gather of the results as the cross-rank consistency check (sweep.gather_results; the GPU ranks send it
    # through RCCL): identical records on both ranks -> consistent; one rank's weights nudged by an ulp -> not
    g_ok = sweep.gather_results(None, res, comm, rccl=False)
    bent = [dict(r) for r in res]
    if comm.rank == 1:
        bent[2] = dict(bent[2], w=np.nextafter(bent[2]["w"], 1.0))
    g_bad = sweep.gather_results(None, bent, comm, rccl=False)
    comm.barrier()
    np.savez(out_path % comm.rank, thetas=np.array([r["theta"] for r in res]),
             fmin=np.array([r["fmin"] for r in res]), ranks=np.array([r["rank"] for r in res]),
             iters=np.array([r["iterations"] for r in res]), w=np.stack([r["w"] for r in res]), tmax=tmax,
             objs=np.array([x for x in comm.allgather_object(comm.rank * 10)]),
             gather_ok=g_ok["consistent"], gather_bad=g_bad["consistent"], gather_ranks=g_ok["ranks"],
             gather_bytes=g_ok["bytes_per_rank"])
    comm.close()
    if mode == "gloo":
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
