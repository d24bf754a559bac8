"""The N > 1 path on the CPU: 2 processes shard a 5-point theta series, gather, and must
reproduce the serial sweep exactly -- once over torch.distributed/gloo (launched the way
the driver launches bench.py) and once over the torch-free SocketComm the GPU ranks use."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden, LBFGS_DEFAULTS
from bioen_amd import sweep
from oracle import oracle_binding as O

WORKER = os.path.join(ROOT, "tests", "_sweep_worker.py")
THETAS = [50.0, 0.5, 5.0, 500.0, 1.0]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def serial_reference():
    d = load_golden("synth_logw_M37xN500.npz")
    out = []
    for th in THETAS:
        g, fmin, code, it, ev = O.opt_lbfgs_logw(d["GInit"], d["G"], d["yTilde"], d["YTilde"], th, LBFGS_DEFAULTS)
        out.append((fmin, it, O.logw_weights(g)[0]))
    return out


def check(tmp_path, tag):
    ref = serial_reference()
    shards = [sweep.shard_thetas(THETAS, r, 2) for r in range(2)]
    for rank in range(2):
        z = np.load(str(tmp_path / ("%s_rank%d.npz" % (tag, rank))))
        assert np.array_equal(z["thetas"], np.array(THETAS))
        assert z["tmax"] == 2.0 and list(z["objs"]) == [0, 10]
        for i, (fmin, it, w) in enumerate(ref):
            assert z["fmin"][i] == fmin and z["iters"][i] == it
            assert np.array_equal(z["w"][i], w)
            assert z["ranks"][i] == (0 if i in shards[0] else 1)


def test_shard_thetas_round_robin_by_cost():
    th = list(np.logspace(3, -0.5, 8))
    for world in (1, 2, 4, 8):
        parts = [sweep.shard_thetas(th, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(8))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    # the two most expensive (smallest) thetas never share a rank when there are >= 2 ranks
    p0, p1 = sweep.shard_thetas(th, 0, 2), sweep.shard_thetas(th, 1, 2)
    assert (7 in p0) != (6 in p0)
    assert sweep.shard_thetas(th, 0, 8) == [7]      # rank 0 gets the smallest theta


def test_single_process_sweep_matches_serial():
    import types
    d = load_golden("synth_logw_M37xN500.npz")

    def solve(theta):
        g, fmin, code, it, ev = O.opt_lbfgs_logw(d["GInit"], d["G"], d["yTilde"], d["YTilde"], theta, LBFGS_DEFAULTS)
        return O.logw_weights(g)[0], types.SimpleNamespace(fmin=fmin, chi2=0.0, kl=0.0, iterations=it,
                                                           evaluations=ev, lbfgs_code=code, seconds=0.0)
    res = sweep.theta_sweep(None, THETAS, solve, n=d["G"].size)
    for r, (fmin, it, w) in zip(res, serial_reference()):
        assert r["fmin"] == fmin and r["iterations"] == it and np.array_equal(r["w"], w) and r["rank"] == 0


@pytest.mark.timeout(300)
def test_two_rank_sweep_gloo(tmp_path):
    port = free_port()
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER, "gloo",
           str(tmp_path / "gloo_rank%d.npz")]
    subprocess.run(cmd, check=True, env=env, timeout=280, cwd=ROOT)
    check(tmp_path, "gloo")


@pytest.mark.timeout(120)
def test_two_rank_sweep_socketcomm(tmp_path):
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TORCHELASTIC_RUN_ID="pytest%d" % os.getpid(), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, WORKER, "socket", str(tmp_path / "sock_rank%d.npz")],
                                      env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=100) == 0
    check(tmp_path, "sock")
