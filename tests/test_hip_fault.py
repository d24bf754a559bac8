"""Failure handling of the multi-rank path: a rank that dies or stalls must turn into an error on
the others within the wait bound -- never a hang.  The reference is a single process and cannot fail this way
(bioen/analyze/procedure.py:62-63); this is the price of the sharded design and is tested as such."""
import json
import os
import signal
import subprocess
import sys
import time

import pytest

from conftest import ROOT
from test_hip_nshard import free_port

pytestmark = pytest.mark.gpu

WORKER = os.path.join(ROOT, "tests", "_fault_worker.py")


def start(tmp_path, mode, extra_env):
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TORCHELASTIC_RUN_ID="fault%d" % os.getpid(), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   **extra_env)
        procs.append(subprocess.Popen([sys.executable, WORKER, str(tmp_path), mode], env=env, cwd=ROOT))
    return procs


def wait_for(path, procs, seconds):
    deadline = time.time() + seconds
    while not os.path.exists(path):
        assert time.time() < deadline, "worker never got ready"
        assert all(p.poll() is None for p in procs), "a worker died before the fault was injected"
        time.sleep(0.05)


@pytest.mark.timeout(300)
def test_killed_rank_turns_into_an_error_on_the_survivor_host_staged(tmp_path):
    procs = start(tmp_path, "kill", {"BIOEN_HIP_WAIT_TIMEOUT": "10"})
    try:
        wait_for(str(tmp_path / "ready0"), procs, 200)
        wait_for(str(tmp_path / "ready1"), procs, 200)
        time.sleep(0.5)                                   # both ranks are solving series now
        t_kill = time.time()
        procs[1].send_signal(signal.SIGKILL)
        assert procs[0].wait(timeout=60) == 0             # the survivor ends by itself, cleanly
        with open(str(tmp_path / "result0.json")) as fp:
            rec = json.load(fp)
        assert rec["transport"] == "host" and rec["first_codes"] and rec["series_completed"] >= 1
        assert rec["error"] and "libbioen_hip" in rec["error"], rec
        assert rec["t_end"] - t_kill < 20.0, rec          # within the bound (a closed socket: at once)
        assert rec["after"] != "usable", rec
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


@pytest.mark.timeout(300)
def test_stalled_rank_times_out_peer_to_peer(tmp_path):
    timeout_s, stall_s = 3.0, 10.0
    procs = start(tmp_path, "stall", {"BIOEN_HIP_WAIT_TIMEOUT": str(timeout_s), "BIOEN_TEST_STALL": str(stall_s)})
    try:
        for p in procs:
            assert p.wait(timeout=200) == 0
        recs = []
        for r in range(2):
            with open(str(tmp_path / ("result%d.json" % r))) as fp:
                recs.append(json.load(fp))
        r0, r1 = recs
        assert r0["transport"] == "p2p" and r1["transport"] == "p2p"
        # rank 0: its exchange kernel gave up after the bound, the host wait reported it -- well before rank 1 woke up
        assert r0["error"] and "peer-to-peer exchange" in r0["error"] and "rank 1" in r0["error"], r0
        assert timeout_s - 0.5 <= r0["t_end"] - r0["t_start"] <= timeout_s + 6.5, r0   # kernel bound, host bound = + 2 s, margin
        assert r0["after"] != "usable"
        # rank 1: woke up into a dead exchange and was told so (ABORT flag or its own bounded wait)
        assert r1["error"] and "peer-to-peer exchange" in r1["error"], r1
        assert r1["t_end"] - r1["t_start"] <= timeout_s + 6.5, r1
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


@pytest.mark.timeout(300)
def test_ranks_out_of_step_get_an_error_not_each_others_numbers(tmp_path):
    """Two ranks that issue DIFFERENT exchanges under the same number (another stage or payload: what ranks do that composed
    different rounds) must both fail loudly and at once: the sender leaves {stage, payload} beside its flag, the receiver
    compares.  (Before r04 the exchange kernel copied whatever the peer had left: garbage in the stage buffer, 1e33 in the
    results of tests/test_hip_nshard.py's early-finisher scenario.)"""
    procs = start(tmp_path, "outofstep", {"BIOEN_HIP_WAIT_TIMEOUT": "20"})
    try:
        for p in procs:
            assert p.wait(timeout=200) == 0
        for r in range(2):
            with open(str(tmp_path / ("result%d.json" % r))) as fp:
                rec = json.load(fp)
            assert rec["transport"] == "p2p" and rec["first_codes"]
            assert rec["error"] and "peer-to-peer exchange" in rec["error"], rec
            assert "OUT OF STEP" in rec["error"] or "ABORT" in rec["error"], rec      # its own comparison, or the peer's verdict
            assert rec["t_end"] - rec["t_start"] < 10.0, rec                          # at flag speed, not at the timeout
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


@pytest.mark.timeout(300)
def test_a_communicator_whose_peer_never_comes_is_given_up(tmp_path):
    """ncclCommInitRank is a collective without a time limit: a rank whose peer never calls it must get an error within three
    times the context's wait bound, not wait for good.  (Own process: the abandoned initialisation stays behind on a helper
    thread.)"""
    script = tmp_path / "lonely.py"
    script.write_text('''
import sys, time
sys.path.insert(0, %r)
import numpy as np
import bioen_amd
ctx = bioen_amd.Context(np.random.default_rng(0).normal(size=(8, 256)), np.zeros(8))
ctx.set_wait_timeout(1.5)
uid = ctx.comm_unique_id()
t0 = time.time()
try:
    ctx.comm_init(uid, 0, 2)                      # rank 0 of 2; rank 1 does not exist
    print("RESULT no error")
except bioen_amd.BioenHipError as e:
    print("RESULT %%.2f %%s" %% (time.time() - t0, e))
f, g = ctx.logw_fdf(np.zeros(256), np.zeros(256), 1.0)          # the context itself is fine without the communicator
print("AFTER", np.isfinite(f))
sys.stdout.flush()
import os
os._exit(0)                                      # (the helper thread is still inside the library's bootstrap)
''' % ROOT)
    out = subprocess.run([sys.executable, str(script)], cwd=ROOT, capture_output=True, text=True, timeout=200)
    lines = [l for l in out.stdout.splitlines() if l.startswith(("RESULT", "AFTER"))]
    assert len(lines) == 2, (out.stdout[-2000:], out.stderr[-2000:])
    res = lines[0].split(None, 2)
    assert res[1] != "no" and "did not return within" in lines[0], lines
    assert 4.0 <= float(res[1]) <= 12.0, lines                  # 3 x 1.5 s, with slack for the library's own start-up
    assert lines[1] == "AFTER True"
