"""Worker of tests/test_hip_fault.py: one rank of a structure-sharded run in which rank 1 goes away.

mode "kill"  (host-staged transport): both ranks solve a theta series again and again; the parent kills rank 1 in the
              middle of one; rank 0 must come back with BioenHipError instead of hanging.
mode "stall" (peer-to-peer transport): rank 1 stops taking part after the first series (it sleeps); rank 0's exchange
              kernel must give up after BIOEN_HIP_WAIT_TIMEOUT, its host wait with it; rank 1, waking up later, must
              fail as well (rank 0's ABORT flag, or its own bounded wait) instead of computing on a dead exchange.
mode "outofstep" (peer-to-peer transport): after a healthy series the ranks issue an exchange of DIFFERENT size under the
              same number (what ranks that composed different rounds would do): both must get an error that says so,
              at flag speed, instead of a stage buffer full of the other rank's numbers."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bioen_amd                      # noqa: E402
from bioen_amd import sweep            # noqa: E402
from conftest import load_golden, LBFGS_DEFAULTS   # noqa: E402


def main():
    out_dir, mode = sys.argv[1], sys.argv[2]
    comm = sweep.SocketComm(timeout=60.0)
    d = load_golden("synth_logw_M64xN2000.npz")
    thetas = [50.0, 5.0, 500.0, 1.0, 20.0]
    ctx = bioen_amd.Context(d["yTilde"], d["YTilde"], device=0, rank=comm.rank, world=comm.world)
    if mode in ("stall", "outofstep"):
        assert sweep.init_p2p(ctx, comm)
    else:
        ctx.set_exchange(comm)
    first = ctx.opt_lbfgs_logw_batch(thetas, d["GInit"], d["G"], LBFGS_DEFAULTS, max_batch=4)   # a healthy series first
    comm.barrier()
    rec = {"rank": comm.rank, "transport": ctx.exchange_transport(), "first_codes": [i.lbfgs_code for i in first[2]]}
    open(os.path.join(out_dir, "ready%d" % comm.rank), "w").close()
    if mode == "stall" and comm.rank == 1:
        time.sleep(float(os.environ.get("BIOEN_TEST_STALL", "12")))
    t0 = time.time()
    err, series = None, 0
    if mode == "outofstep":
        try:
            ctx.exchange_probe(count=64 + comm.rank, reps=3)
        except bioen_amd.BioenHipError as e:
            err = str(e)
        rec.update(error=err, series_completed=0, t_start=t0, t_end=time.time())
        with open(os.path.join(out_dir, "result%d.json" % comm.rank), "w") as fp:
            json.dump(rec, fp)
        deadline = time.time() + 30          # stay mapped until the peer has its verdict too
        while not os.path.exists(os.path.join(out_dir, "result%d.json" % (1 - comm.rank))) and time.time() < deadline:
            time.sleep(0.05)
        ctx.close()
        return
    try:
        for _ in range(2000 if mode == "kill" else 1):
            ctx.opt_lbfgs_logw_batch(thetas, d["GInit"], d["G"], LBFGS_DEFAULTS, max_batch=4)
            series += 1
    except bioen_amd.BioenHipError as e:
        err = str(e)
    rec.update(error=err, series_completed=series, t_start=t0, t_end=time.time())
    try:                                   # the context is unusable after a failed wait ...
        ctx.logw_weights(d["GInit"])
        rec["after"] = "usable"
    except bioen_amd.BioenHipError as e:
        rec["after"] = str(e)
    with open(os.path.join(out_dir, "result%d.json" % comm.rank), "w") as fp:
        json.dump(rec, fp)
    if mode == "stall" and comm.rank == 0:   # ... and stays mapped until the peer has seen the failure too
        deadline = time.time() + 60
        while not os.path.exists(os.path.join(out_dir, "result1.json")) and time.time() < deadline:
            time.sleep(0.1)
    ctx.close()                            # must not hang either


if __name__ == "__main__":
    main()
