"""Worker of tests/test_hip_nshard.py: one rank of a structure-sharded (column-sharded) run.
All ranks share the single GPU of the test box, so the cross-rank all-gathers go through the
host-staged exchange hook (SocketComm) or -- BIOEN_TEST_TRANSPORT=p2p -- through the peer-to-peer
mailboxes (hipIpc works between processes on one GPU); on a real multi-GPU node the same code path
uses the mailboxes over xGMI, or RCCL."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bioen_amd                      # noqa: E402
from bioen_amd import sweep            # noqa: E402
from conftest import load_golden, tall_forces_problem, LBFGS_DEFAULTS, LBFGS_CONV   # noqa: E402


def main():
    out_path = sys.argv[1]
    comm = sweep.SocketComm()
    d = load_golden("synth_logw_M64xN2000.npz")
    thetas = [50.0, 5.0, 500.0, 1.0, 20.0]
    rng = np.random.default_rng(99)
    g = d["GInit"].ravel() + 0.2 * rng.standard_normal(d["GInit"].size)

    transport = os.environ.get("BIOEN_TEST_TRANSPORT", "host")

    def attach(c):
        if transport == "p2p":
            assert sweep.init_p2p(c, comm), "the peer-to-peer exchange did not attach"
            assert c.exchange_transport() == "p2p"
        else:
            c.set_exchange(comm)
            assert c.exchange_transport() == "host"

    ctx = bioen_amd.Context(d["yTilde"], d["YTilde"], device=0, rank=comm.rank, world=comm.world)
    attach(ctx)
    w, logs = ctx.logw_weights(g)
    f, grad = ctx.logw_fdf(g, d["G"], d["theta"])
    res, wopt, infos = ctx.opt_lbfgs_logw_batch(thetas, d["GInit"], d["G"], LBFGS_DEFAULTS, max_batch=4)
    spec = ctx.speculation_stats()                 # sharded contexts shadow the slowest thetas' line searches by default
    def same(x, y):
        return bool(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and
                    [(i.fmin, i.iterations, i.evaluations, i.lbfgs_code) for i in x[2]] ==
                    [(i.fmin, i.iterations, i.evaluations, i.lbfgs_code) for i in y[2]])
    # ... and must land on the same bits without them, and with the late Gram pass (a third all-gather) instead of the
    # shadows' own sweep
    os.environ["BIOEN_HIP_SHADOWS"] = "0"
    plain = ctx.opt_lbfgs_logw_batch(thetas, d["GInit"], d["G"], LBFGS_DEFAULTS, max_batch=4)
    del os.environ["BIOEN_HIP_SHADOWS"]
    os.environ["BIOEN_HIP_SHADOW_GRAM"] = "0"
    late = ctx.opt_lbfgs_logw_batch(thetas, d["GInit"], d["G"], LBFGS_DEFAULTS, max_batch=4)
    del os.environ["BIOEN_HIP_SHADOW_GRAM"]
    same_without = same((res, wopt, infos), plain) and same((res, wopt, infos), late)
    # a series that fills the batch: two slots are kept back for the shadows (the headline's shape at 8 GPUs)
    th8 = [300.0, 100.0, 30.0, 10.0, 3.0, 1.0, 0.3, 0.1]
    full = ctx.opt_lbfgs_logw_batch(th8, d["GInit"], d["G"], LBFGS_DEFAULTS, max_batch=8)
    os.environ["BIOEN_HIP_SHADOWS"] = "0"
    full_plain = ctx.opt_lbfgs_logw_batch(th8, d["GInit"], d["G"], LBFGS_DEFAULTS, max_batch=8)
    del os.environ["BIOEN_HIP_SHADOWS"]
    same_without = same_without and same(full, full_plain)
    # the converged run the reference's golden pins (tests/golden: lbfgs_conv_*)
    gconv, wconv, iconv = ctx.opt_lbfgs_logw(d["GInit"], d["G"], d["theta"], LBFGS_CONV)
    # a non-finite start in ONE rank's block (the last structure): every rank sees it after the exchange and ends the run
    # the way the reference's binary does (status 2, one evaluation); a NaN theta ends alone inside a batch
    g_nan = d["GInit"].ravel().copy()
    g_nan[-1] = np.nan
    _, _, inan = ctx.opt_lbfgs_logw(g_nan, d["G"], d["theta"], LBFGS_DEFAULTS)
    nanb = ctx.opt_lbfgs_logw_batch([50.0, float("nan"), 5.0], d["GInit"], d["G"], LBFGS_DEFAULTS, max_batch=4)
    # ... and in the order in which the NaN problem takes the first slot and leaves it in the first round, while the
    # shadows of the others want a slot: every rank must compose the same rounds whatever its delivery thread is doing
    nanc = ctx.opt_lbfgs_logw_batch([float("nan"), 50.0, 5.0], d["GInit"], d["G"], LBFGS_DEFAULTS, max_batch=4)
    assert [(i.fmin, i.iterations) for i in nanc[2][1:]] == [(i.fmin, i.iterations) for i in (nanb[2][0], nanb[2][2])]
    assert np.array_equal(nanc[0][1:], nanb[0][[0, 2]])
    # The smallest theta of a batch -- the one the shadows work for -- leaves in its first round (it starts at its own
    # optimum) while the others run on: its shadows' slots rest for a few rounds, and the slot it has just freed is the
    # only one a new shadow can take.  Whether that slot's result delivery has completed is a matter of thread timing on
    # each rank; the rounds the ranks compose must not depend on it (test_a_dawdling_rank_changes_no_bit).
    gopt = ctx.opt_lbfgs_logw(d["GInit"], d["G"], 0.5, LBFGS_CONV)[0]
    starts = np.stack([gopt, d["GInit"].ravel(), d["GInit"].ravel()])
    early = ctx.opt_lbfgs_logw_batch([0.5, 50.0, 5.0], starts, d["G"], LBFGS_DEFAULTS, max_batch=4)
    chi2, yave = ctx.chi_squared(w)
    # r05: the GSL-style minimizers on a sharded context -- their inner products and norms are sums over
    # structures like every other one (canonical segments, one stage all-gather each)
    ggsl, wgsl, igsl = ctx.opt_gsl_logw(d["GInit"], d["G"], d["theta"], "bfgs2", dict(step_size=0.01, tol=1e-3, max_iterations=200))
    gcg, _, icg = ctx.opt_gsl_logw(d["GInit"], d["G"], d["theta"], "conjugate_pr", dict(step_size=0.01, tol=1e-3, max_iterations=60),
                                   want_weights=False)
    block = ctx.read_ytilde()
    col0, n_local = ctx.col0, ctx.n_local
    counts = ctx.exchange_counts3()
    probe_us = ctx.exchange_probe(count=64 * 8, reps=200)
    # transports come and go on a live context (bench.py measures them and detaches the slower one): the same series over
    # the host-staged path after a detach, and over the mailboxes again after a second attach, returns the same bits
    if transport == "p2p":
        ctx.p2p_detach()
        ctx.set_exchange(comm)
        assert ctx.exchange_transport() == "host"
        again_host = ctx.opt_lbfgs_logw_batch(thetas, d["GInit"], d["G"], LBFGS_DEFAULTS, max_batch=4)
        assert sweep.init_p2p(ctx, comm) and ctx.exchange_transport() == "p2p"
        again_p2p = ctx.opt_lbfgs_logw_batch(thetas, d["GInit"], d["G"], LBFGS_DEFAULTS, max_batch=4)
        assert same((res, wopt, infos), again_host) and same((res, wopt, infos), again_p2p)
    ctx.close()

    # forces method on a sharded context (strip passes: 2 all-gathers per evaluation)
    fd = load_golden("synth_forces_M96xN3000.npz")
    fctx = bioen_amd.Context(fd["yTilde"], fd["YTilde"], device=0, rank=comm.rank, world=comm.world)
    attach(fctx)
    f0 = 1e-3 * np.random.default_rng(5).standard_normal(fd["yTilde"].shape[0])
    fwts = fctx.forces_weights(f0, fd["w0"])           # r05: served on sharded contexts (pass 1 of the strip evaluation)
    ff, fgrad = fctx.forces_fdf(f0, fd["w0"], 10.0)
    fthetas = [100.0, 10.0, 1000.0]
    fres, fw, finfos = fctx.opt_lbfgs_forces_batch(fthetas, fd["forces_init"], fd["w0"], LBFGS_DEFAULTS)
    fconv, fwconv, ficonv = fctx.opt_lbfgs_forces(fd["forces_init"], fd["w0"], fd["theta"], LBFGS_CONV)
    # a batch wider than four: the strip kernel's K > 4 form (row-sum product deferred behind the next strip's barrier)
    f6res, f6w, f6infos = fctx.opt_lbfgs_forces_batch([300.0, 100.0, 30.0, 10.0, 3.0, 1.0], fd["forces_init"], fd["w0"],
                                                      LBFGS_DEFAULTS, max_batch=6)
    fctx.close()
    # r05: more than 1024 observables -- the four passes over row panels, in canonical segments like the two strip passes
    td = tall_forces_problem()
    tctx = bioen_amd.Context(td["yTilde"], td["YTilde"], device=0, rank=comm.rank, world=comm.world)
    attach(tctx)
    tw = tctx.forces_weights(td["f0"], td["w0"])
    tf, tgrad = tctx.forces_fdf(td["f0"], td["w0"], 100.0)
    tres, tww, tinfos = tctx.opt_lbfgs_forces_batch(td["thetas"], np.zeros(td["f0"].size), td["w0"],
                                                    dict(LBFGS_DEFAULTS, max_iterations=25))
    tctx.close()
    comm.barrier()
    np.savez(out_path % comm.rank, w=w, logs=logs, f=f, grad=grad, res=res, wopt=wopt,
             fmin=np.array([i.fmin for i in infos]), iters=np.array([i.iterations for i in infos]),
             evals=np.array([i.evaluations for i in infos]), codes=np.array([i.lbfgs_code for i in infos]),
             chi2=np.array([i.chi2 for i in infos]), kl=np.array([i.kl for i in infos]),
             block=block, col0=col0, n_local=n_local, chi2w=chi2, yave=yave,
             ff=ff, fgrad=fgrad, fres=fres, fw=fw, ffmin=np.array([i.fmin for i in finfos]),
             fiters=np.array([i.iterations for i in finfos]), fcodes=np.array([i.lbfgs_code for i in finfos]),
             fkl=np.array([i.kl for i in finfos]), fchi2=np.array([i.chi2 for i in finfos]),
             spec=np.array(spec), same_without=same_without, counts=np.array(counts), probe_us=probe_us,
             wconv=wconv, fminconv=iconv.fmin, codeconv=iconv.lbfgs_code,
             fwconv=fwconv, ffminconv=ficonv.fmin, fcodeconv=ficonv.lbfgs_code,
             f6res=f6res, f6w=f6w, f6fmin=np.array([i.fmin for i in f6infos]),
             nan_code=inan.lbfgs_code, nan_evals=inan.evaluations, nanb_codes=np.array([i.lbfgs_code for i in nanb[2]]),
             nanb_evals=np.array([i.evaluations for i in nanb[2]]), nanb_fmin=np.array([i.fmin for i in nanb[2]]),
             nanb_res=nanb[0][[0, 2]], early_res=early[0], early_fmin=np.array([i.fmin for i in early[2]]),
             early_codes=np.array([i.lbfgs_code for i in early[2]]), early_evals=np.array([i.evaluations for i in early[2]]),
             ggsl=ggsl, wgsl=wgsl, gsl_stat=np.array([igsl.fmin, igsl.lbfgs_code, igsl.iterations, igsl.evaluations]),
             gcg=gcg, cg_stat=np.array([icg.fmin, icg.lbfgs_code, icg.iterations, icg.evaluations]), fwts=fwts,
             tw=tw, tf=tf, tgrad=tgrad, tres=tres, tww=tww,
             tstat=np.array([(i.fmin, i.iterations, i.evaluations, i.lbfgs_code) for i in tinfos]))
    comm.close()


if __name__ == "__main__":
    main()
