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
