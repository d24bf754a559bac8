"""GPU tests of the RCCL STAGE path on one GPU.

On 8 GPUs every reduction over structures is completed by an in-place ncclAllGather of an exchange stage, issued
on the context's stream between the kernels that fill and consume it (api.hip: exchange).  A one-GPU box cannot
run several RCCL ranks, but it can run the SAME call sequence with one rank: `set_force_exchange` makes an unsharded
context with a 1-rank communicator execute every stage all-gather of the sharded code path (log-weights: ybar +
softmax totals, gradient dot products, Gram products; forces: ybar, gradient shares).  The all-gather of one rank
copies the segment onto itself, so every result must equal the plain context's to the last bit -- and the
exchange counters must show that the collectives really ran through RCCL."""
import numpy as np
import pytest

from conftest import load_golden, LBFGS_DEFAULTS, LBFGS_CONV

pytestmark = pytest.mark.gpu


def _rccl_ctx(bioen_amd, yTilde, YTilde):
    ctx = bioen_amd.Context(yTilde, YTilde)
    try:
        ctx.comm_init(bioen_amd.Context.comm_unique_id(), 0, 1)
    except bioen_amd.BioenHipError as e:            # pragma: no cover - the image ships librccl
        ctx.close()
        pytest.skip("RCCL cannot initialise here: %s" % e)
    ctx.set_force_exchange(True)
    return ctx


def _problem():
    d = load_golden("synth_logw_M64xN2000.npz")
    rng = np.random.default_rng(17)
    g0 = d["GInit"].ravel() + 0.1 * rng.standard_normal(d["GInit"].size)
    return d, g0


@pytest.mark.parametrize("params", [LBFGS_DEFAULTS, LBFGS_CONV, dict(LBFGS_DEFAULTS, linesearch=0)],
                         ids=["yaml", "converged", "more-thuente"])
def test_logw_series_through_rccl_stage_exchanges_is_bitwise_the_plain_run(params):
    import bioen_amd
    d, g0 = _problem()
    thetas = [300.0, 30.0, 3.0, 0.3, 100.0]
    with bioen_amd.Context(d["yTilde"], d["YTilde"]) as plain:
        res0, w0, info0 = plain.opt_lbfgs_logw_batch(thetas, g0, d["G"], params, max_batch=4)
        f0, grad0 = plain.logw_fdf(g0, d["G"], 10.0)
        assert plain.exchange_counts() == (0, 0)
    ctx = _rccl_ctx(bioen_amd, d["yTilde"], d["YTilde"])
    try:
        res1, w1, info1 = ctx.opt_lbfgs_logw_batch(thetas, g0, d["G"], params, max_batch=4)
        f1, grad1 = ctx.logw_fdf(g0, d["G"], 10.0)
        n_rccl, n_host = ctx.exchange_counts()
    finally:
        ctx.close()
    rounds = sum(i.evaluations for i in info1) // 4          # at least this many lock-step rounds
    assert n_host == 0 and n_rccl >= 2 * rounds, (n_rccl, rounds)
    assert np.array_equal(res0, res1) and np.array_equal(w0, w1)
    assert f0 == f1 and np.array_equal(grad0, grad1)
    for a, b in zip(info0, info1):
        assert (a.fmin, a.chi2, a.kl, a.lbfgs_code, a.iterations, a.evaluations) == \
               (b.fmin, b.chi2, b.kl, b.lbfgs_code, b.iterations, b.evaluations)


def test_logw_two_loop_direction_through_rccl_stage_exchanges():
    """liblbfgs' literal order of operations needs 2 bound + 2 exchanges per direction (X_SY, X_REC0/1, X_DGI)"""
    import bioen_amd
    d, g0 = _problem()
    with bioen_amd.Context(d["yTilde"], d["YTilde"]) as plain:
        plain.set_direction_mode("twoloop")
        r0, w0, i0 = plain.opt_lbfgs_logw(g0, d["G"], 20.0, LBFGS_DEFAULTS)
    ctx = _rccl_ctx(bioen_amd, d["yTilde"], d["YTilde"])
    try:
        ctx.set_direction_mode("twoloop")
        r1, w1, i1 = ctx.opt_lbfgs_logw(g0, d["G"], 20.0, LBFGS_DEFAULTS)
        n_rccl, _ = ctx.exchange_counts()
    finally:
        ctx.close()
    assert n_rccl > 10 * i1.iterations
    assert np.array_equal(r0, r1) and np.array_equal(w0, w1) and i0.fmin == i1.fmin and i0.iterations == i1.iterations


def test_forces_series_through_rccl_stage_exchanges_is_bitwise_the_plain_run():
    import bioen_amd
    d = load_golden("synth_forces_M96xN3000.npz")
    thetas = [100.0, 10.0, 1.0]
    f0 = np.zeros(d["yTilde"].shape[0])
    w0 = d["w0"].ravel()
    with bioen_amd.Context(d["yTilde"], d["YTilde"]) as plain:
        res0, wa, info0 = plain.opt_lbfgs_forces_batch(thetas, f0, w0, LBFGS_DEFAULTS, max_batch=3)
        ff0, fg0 = plain.forces_fdf(res0[0], w0, thetas[0])
    ctx = _rccl_ctx(bioen_amd, d["yTilde"], d["YTilde"])
    try:
        res1, wb, info1 = ctx.opt_lbfgs_forces_batch(thetas, f0, w0, LBFGS_DEFAULTS, max_batch=3)
        ff1, fg1 = ctx.forces_fdf(res0[0], w0, thetas[0])
        n_rccl, n_host = ctx.exchange_counts()
        adopted = ctx.speculation_stats()[1]          # evaluations that cost no round of their own (shadows in free slots)
    finally:
        ctx.close()
    assert n_host == 0 and n_rccl >= 2 * (max(i.evaluations for i in info1) - adopted)
    assert np.array_equal(res0, res1) and np.array_equal(wa, wb)
    assert ff0 == ff1 and np.array_equal(fg0, fg1)
    for a, b in zip(info0, info1):
        assert (a.fmin, a.lbfgs_code, a.iterations, a.evaluations) == (b.fmin, b.lbfgs_code, b.iterations, b.evaluations)


def test_comm_allgather_and_exchange_probe_on_one_rank():
    import bioen_amd
    d, _ = _problem()
    ctx = _rccl_ctx(bioen_amd, d["yTilde"], d["YTilde"])
    try:
        send = np.arange(1000, dtype=np.float64) * 0.5
        got = ctx.comm_allgather(send, 1)
        assert got.shape == (1, 1000) and np.array_equal(got[0], send)
        before = ctx.exchange_counts()[0]
        us = ctx.exchange_probe(count=64 * 8, reps=20)
        assert ctx.exchange_counts()[0] - before == 25        # 5 warm-up + 20 timed all-gathers
        assert 0.0 < us < 5e4
    finally:
        ctx.close()


def test_host_callback_exchange_on_one_rank_is_bitwise_too():
    """the host-staged transport of the same stage path (what the multi-process tests on one GPU use)"""
    import bioen_amd

    class OneRank(object):
        calls = 0

        def allgather_array(self, a):
            OneRank.calls += 1
            return a[None, :]

    d, g0 = _problem()
    with bioen_amd.Context(d["yTilde"], d["YTilde"]) as plain:
        r0, w0, i0 = plain.opt_lbfgs_logw(g0, d["G"], 20.0, LBFGS_DEFAULTS)
    with bioen_amd.Context(d["yTilde"], d["YTilde"]) as ctx:
        ctx.set_exchange(OneRank())
        ctx.set_force_exchange(True)
        r1, w1, i1 = ctx.opt_lbfgs_logw(g0, d["G"], 20.0, LBFGS_DEFAULTS)
        assert ctx.exchange_counts()[1] == OneRank.calls > 2 * i1.iterations
    assert np.array_equal(r0, r1) and np.array_equal(w0, w1) and i0.fmin == i1.fmin


def test_peer_to_peer_transport_on_one_rank_is_bitwise_too():
    """the third transport of the stage path with nobody to exchange with: the (empty) exchange kernels are launched at
    every stage, the self-test runs, nothing moves by a bit"""
    import bioen_amd
    d, g0 = _problem()
    with bioen_amd.Context(d["yTilde"], d["YTilde"]) as plain:
        r0, w0, i0 = plain.opt_lbfgs_logw(g0, d["G"], 20.0, LBFGS_DEFAULTS)
    with bioen_amd.Context(d["yTilde"], d["YTilde"]) as ctx:
        assert ctx.exchange_transport() == "none"
        assert len(ctx.p2p_export()) == 64
        ctx.p2p_attach(None)
        ctx.set_force_exchange(True)
        assert ctx.exchange_transport() == "p2p"
        assert ctx.exchange_selftest(20) == 0
        r1, w1, i1 = ctx.opt_lbfgs_logw(g0, d["G"], 20.0, LBFGS_DEFAULTS)
        assert ctx.exchange_counts3()[2] > 2 * i1.iterations and ctx.exchange_counts3()[:2] == (0, 0)
        ctx.p2p_detach()
        assert ctx.exchange_transport() == "none"
    assert np.array_equal(r0, r1) and np.array_equal(w0, w1) and i0.fmin == i1.fmin
