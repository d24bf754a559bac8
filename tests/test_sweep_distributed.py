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
        # sweep.gather_results: one all-gather of every theta's (scalars, weights); consistent only if all ranks hold the same bytes
        assert bool(z["gather_ok"]) and not bool(z["gather_bad"]) and int(z["gather_ranks"]) == 2
        assert int(z["gather_bytes"]) == len(THETAS) * (sweep.HEADER + 500) * 8
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


def test_socketcomm_control_messages_are_not_pickled():
    """Control objects travel as JSON (bytes hex-wrapped); anything else is refused at the sender and a
    forged payload at the receiver -- nothing from the network is unpickled."""
    msg = (b"\x00\x01" * 64, None, True, 1.5, float("inf"), "text", [1, 2])
    back = sweep._dec(sweep._enc(msg))
    assert back == [b"\x00\x01" * 64, None, True, 1.5, float("inf"), "text", [1, 2]]
    with pytest.raises(TypeError):
        sweep._enc(object())
    with pytest.raises(ValueError):
        sweep._dec(b'{"__reduce__": "os.system"}')
    import inspect
    assert "pickle" not in inspect.getsource(sweep).replace("unpickled", "").replace("no pickle", "")


@pytest.mark.timeout(60)
def test_socketcomm_refuses_a_peer_without_the_token(tmp_path):
    """Rank 0 keeps waiting for the real rank 1 when a stranger that knows the port and the magic
    (but not the 32-byte token of the private rendezvous file) connects first."""
    import struct
    import threading
    import time
    port = free_port()
    run_id = "tok%d" % os.getpid()
    env0 = dict(os.environ, RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                TORCHELASTIC_RUN_ID=run_id)
    code = ("import sys; sys.path.insert(0, %r); from bioen_amd import sweep; c = sweep.SocketComm(timeout=40);"
            "print(c.allgather_object(c.rank)); c.close()" % ROOT)
    p0 = subprocess.Popen([sys.executable, "-c", code], env=env0, stdout=subprocess.PIPE, text=True)
    path = os.path.join("/tmp", "bioen_amd_%d" % os.getuid(), "rdzv_%d_%s" % (port, run_id))
    for _ in range(200):
        if os.path.exists(path):
            break
        time.sleep(0.05)
    st = os.stat(path)
    assert st.st_mode & 0o077 == 0 and os.stat(os.path.dirname(path)).st_mode & 0o077 == 0
    listen_port = int(open(path).read().split()[0])
    s = socket.create_connection(("127.0.0.1", listen_port), timeout=5)
    s.sendall(sweep.SocketComm.MAGIC + struct.pack("<i", 1) + b"\x00" * 32)      # right magic, wrong token
    s.settimeout(5)
    try:
        assert s.recv(8) == b""          # closed without the acknowledgement
    except (ConnectionResetError, socket.timeout):
        pass
    s.close()
    env1 = dict(env0, RANK="1")
    p1 = subprocess.Popen([sys.executable, "-c", code], env=env1, stdout=subprocess.PIPE, text=True)
    assert p1.wait(timeout=45) == 0 and p0.wait(timeout=45) == 0
    assert p0.stdout.read().strip() == "[0, 1]" and p1.stdout.read().strip() == "[0, 1]"


@pytest.mark.timeout(120)
@pytest.mark.parametrize("fault", ["none", "export", "attach", "selftest"])
def test_init_p2p_outcome_is_agreed_between_ranks(tmp_path, fault):
    """sweep.init_p2p over two real SocketComm ranks with FAKE contexts (no GPU): the handles are all-gathered in rank
    order, and whenever ONE rank cannot export, map or pass the self-test, BOTH return False and BOTH detach -- a
    transport half attached would hang the first exchange."""
    port = free_port()
    code = r'''
import sys, json
sys.path.insert(0, %r)
from bioen_amd import sweep, BioenHipError
fault, out = sys.argv[1], sys.argv[2]
comm = sweep.SocketComm(timeout=40)
class Ctx(object):
    world = 2
    def __init__(self): self.log = []
    def p2p_export(self):
        if fault == "export" and comm.rank == 1: raise BioenHipError("no handle here")
        return bytes([comm.rank]) * 64
    def p2p_attach(self, handles):
        self.log.append(("attach", [h[0] for h in handles]))
        if fault == "attach" and comm.rank == 0: raise BioenHipError("cannot map")
    def exchange_selftest(self, reps):
        self.log.append(("selftest", reps))
        return 3 if (fault == "selftest" and comm.rank == 1) else 0
    def p2p_detach(self): self.log.append(("detach",))
ctx = Ctx()
ok = sweep.init_p2p(ctx, comm)
json.dump({"ok": ok, "log": ctx.log}, open(out %% comm.rank, "w"))
comm.close()
''' % ROOT
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TORCHELASTIC_RUN_ID="p2pagree%d%s" % (os.getpid(), fault))
        procs.append(subprocess.Popen([sys.executable, "-c", code, fault, str(tmp_path / "r%d.json")], env=env))
    for p in procs:
        assert p.wait(timeout=100) == 0
    import json
    recs = [json.load(open(str(tmp_path / ("r%d.json" % r)))) for r in range(2)]
    assert recs[0]["ok"] == recs[1]["ok"] == (fault == "none")
    for r in recs:
        kinds = [e[0] for e in r["log"]]
        if fault == "none":
            assert kinds == ["attach", "selftest"] and r["log"][0][1] == [0, 1]      # handles in rank order
        else:
            assert kinds[-1] == "detach"                                              # every rank lets go
            assert ("attach" in kinds) == (fault != "export")
            assert ("selftest" in kinds) == (fault == "selftest")


def test_thread_comm_is_an_allgather_between_threads():
    """sweep.ThreadComm: ranks as threads of one process (the 8-rank GPU test's control plane and host-staged transport)"""
    import threading
    comms = sweep.ThreadComm.create(3)
    out = [None] * 3

    def rank_main(r):
        a = comms[r].allgather_array(np.arange(4.0) + 10 * r)
        b = comms[r].allgather_array(np.full(2, float(r)))          # back to back: the slots are reused safely
        out[r] = (a, b, comms[r].max(r + 0.5), comms[r].allgather_object({"rank": r}))
        comms[r].barrier()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(3)]
    [t.start() for t in ts]
    [t.join(timeout=30) for t in ts]
    for r in range(3):
        a, b, mx, objs = out[r]
        assert a.shape == (3, 4) and np.array_equal(a[:, 0], [0.0, 10.0, 20.0]) and np.array_equal(b[:, 1], [0.0, 1.0, 2.0])
        assert mx == 2.5 and [o["rank"] for o in objs] == [0, 1, 2] and comms[r].rank == r and comms[r].world == 3
