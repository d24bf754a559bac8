"""bench.py's host-side helpers that need no GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def test_pmc_counter_csv_is_reduced_to_per_launch_means(tmp_path):
    """rocprofv3 --pmc writes one row per launch and counter; the bench wants the mean per kernel BASE name (all batch
    widths of a template together), and only of the counter asked for."""
    d = tmp_path / "host" / "123"
    d.mkdir(parents=True)
    rows = ["Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,"
            "Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,"
            "Start_Timestamp,End_Timestamp"]
    def row(name, ctr, val):
        return '1,1,1,1,1,1,1024,7,"%s",256,0,0,64,0,32,%s,%s,0,1' % (name, ctr, val)
    rows += [row("void bioen::k_strip_adj<8, true>(bioen::StripArgs, bioen::MVec8, bioen::MVec8)", "FETCH_SIZE", 4000000.0),
             row("void bioen::k_strip_adj<1, true>(bioen::StripArgs, bioen::MVec8, bioen::MVec8)", "FETCH_SIZE", 4200000.0),
             row("void bioen::k_strip_adj<8, true>(bioen::StripArgs, bioen::MVec8, bioen::MVec8)", "SQ_WAVES", 7.0),
             row("void bioen::k_strip_fwd<8, true>(bioen::StripArgs, bioen::Vec8)", "FETCH_SIZE", 1.0),
             row("bioen::k_gram(bioen::GramArgs, int, bioen::Xch)", "FETCH_SIZE", 5.0)]
    (d / "pmc_counter_collection.csv").write_text("\n".join(rows) + "\n")
    means = bench.pmc_kernel_means(str(tmp_path), "FETCH_SIZE")
    assert means["k_strip_adj"] == (4100000.0, 2)
    assert means["k_strip_fwd"] == (1.0, 1) and means["k_gram"] == (5.0, 1)
    assert bench.pmc_kernel_means(str(tmp_path), "WRITE_SIZE") == {}


def test_survey_inputs_follow_the_recipe_stream():
    """ONE default_rng(12345) stream -- YTrue, the matrix row by row, then the targets."""
    M, N = 5, 40
    y, Y = bench.survey_inputs(M, N, seed=12345)
    rng = np.random.default_rng(12345)
    YTrue = rng.uniform(1, 10, M)
    first_row = rng.normal(YTrue[0], 0.5 * YTrue[0], N) / (0.1 * YTrue[0])
    assert y.shape == (M, N) and Y.shape == (M,) and np.array_equal(y[0], first_row)
    assert abs(y[3].mean() - 10.0) < 5.0                      # ytilde_i ~ N(10, 5): mean YTrue_i / (0.1 YTrue_i)


def test_profiler_detection_reads_the_environment(monkeypatch):
    monkeypatch.delenv("ROCP_TOOL_LIBRARIES", raising=False)
    monkeypatch.setenv("LD_PRELOAD", "/some/guard.so")
    monkeypatch.delenv("HSA_TOOLS_LIB", raising=False)
    assert not bench.under_profiler()
    monkeypatch.setenv("LD_PRELOAD", "/some/guard.so:/opt/rocm-7.2.0/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.under_profiler()


# ---- which transport a multi-rank run ends up on (bench.choose_transport), with fakes of the context and the control plane ----
class _Err(Exception):
    pass


class _Comm(object):
    """one rank's view of a control plane whose other ranks answer `others(x)` to an all-gather of x"""
    def __init__(self, others=lambda x: [x]):
        self.others = others

    def allgather_object(self, x):
        return [x] + list(self.others(x))


class _Ctx(object):
    def __init__(self, us):
        self.us, self.log, self.p2p, self.rccl, self.host = us, [], False, False, False

    def exchange_transport(self):
        return "p2p" if self.p2p else "rccl" if self.rccl else "host" if self.host else "none"

    def exchange_probe(self, count, reps):
        t = self.us[self.exchange_transport()]
        if isinstance(t, Exception):
            raise t
        return t

    def p2p_detach(self):
        self.log.append("p2p_detach"); self.p2p = False

    def comm_destroy(self):
        self.log.append("comm_destroy"); self.rccl = False

    def set_exchange(self, comm):
        self.log.append("set_exchange"); self.host = True


class _Sweep(object):
    def __init__(self, p2p_ok=True, rccl_ok=True):
        self.p2p_ok, self.rccl_ok, self.calls = p2p_ok, rccl_ok, []

    def init_p2p(self, ctx, comm):
        self.calls.append("init_p2p")
        ctx.p2p = bool(self.p2p_ok)
        return ctx.p2p

    def init_rccl(self, ctx, comm):
        self.calls.append("init_rccl")
        if isinstance(self.rccl_ok, Exception):
            raise self.rccl_ok
        ctx.rccl = bool(self.rccl_ok)
        return ctx.rccl


class _Quiet(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _choose(transport, sweep, us, nshard=True, others=lambda x: [x]):
    ctx, xinfo = _Ctx(us), {}
    gather, rccl = bench.choose_transport(ctx, _Comm(others), sweep, nshard, transport, 8192, xinfo, _Err, quiet=_Quiet)
    return ctx, sweep, xinfo, gather, rccl


def test_auto_takes_the_mailboxes_for_the_rounds_and_rccl_for_the_final_gather():
    """r05: every multi-rank run owns an RCCL communicator over all ranks for the final all-gather of the results; the
    mailboxes keep the in-loop stage exchanges (the library prefers them)"""
    ctx, sw, x, gather, rccl = _choose("auto", _Sweep(), {"p2p": 5.0, "rccl": 25.0, "host": 90.0})
    assert sw.calls == ["init_p2p", "init_rccl"] and ctx.log == [] and rccl and x["rccl_gather"] is True
    assert x["transport"] == "p2p" and x["exchange_us"] == 5.0 and "rccl_us" not in x and "stage exchanges" in gather
    assert "rccl-allgather of the results" in gather and "rccl_error" not in x


def test_rccl_down_beside_working_mailboxes_is_reported_not_fatal():
    ctx, sw, x, gather, rccl = _choose("auto", _Sweep(rccl_ok=_Err("ncclCommInitRank: invalid usage")), {"p2p": 5.0, "rccl": 25.0, "host": 90.0})
    assert sw.calls == ["init_p2p", "init_rccl"] and not rccl and x["rccl_gather"] is False and x["transport"] == "p2p"
    assert "invalid usage" in x["rccl_error"] and ctx.log == []
    # ... and when another rank could not: this rank gives its communicator up again, the mailboxes stay
    others = lambda v: [False] if isinstance(v, bool) else [v]
    ctx, sw, x, gather, rccl = _choose("auto", _Sweep(), {"p2p": 5.0, "rccl": 25.0, "host": 90.0}, others=others)
    assert not rccl and ctx.log == ["comm_destroy", "set_exchange"] or (not rccl and "comm_destroy" in ctx.log)
    assert x["rccl_gather"] is False and "failed on some rank" in x["rccl_error"]


def test_auto_falls_back_to_rccl_then_to_the_host():
    ctx, sw, x, gather, rccl = _choose("auto", _Sweep(p2p_ok=False), {"p2p": 5.0, "rccl": 25.0, "host": 90.0})
    assert sw.calls == ["init_p2p", "init_rccl"] and rccl and x["transport"] == "rccl" and x["exchange_us"] == 25.0
    assert gather == "rccl-allgather" and x["p2p_attached"] is False and x["rccl_gather"] is True
    ctx, sw, x, gather, rccl = _choose("auto", _Sweep(p2p_ok=False, rccl_ok=_Err("no librccl")), {"p2p": 5.0, "rccl": 25.0, "host": 90.0})
    assert not rccl and x["transport"] == "host" and x["exchange_us"] == 90.0 and ctx.log == ["set_exchange"]
    assert "RCCL unavailable" in gather and "RCCL unavailable" in x["rccl_error"] and sw.calls == ["init_p2p", "init_rccl"]


def test_forced_host_transport_leaves_rccl_alone():
    ctx, sw, x, gather, rccl = _choose("host", _Sweep(), {"p2p": 5.0, "rccl": 25.0, "host": 90.0})
    assert sw.calls == [] and not rccl and x["transport"] == "host" and x["rccl_gather"] is False and "host" in x["rccl_error"]


def test_rccl_that_one_rank_could_not_initialise_is_given_up_by_all():
    # this rank initialised its communicator, another one reports False: destroy it, host-staged for everybody
    others = lambda v: [False] if isinstance(v, bool) else [v]
    ctx, sw, x, gather, rccl = _choose("rccl", _Sweep(), {"p2p": 5.0, "rccl": 25.0, "host": 90.0}, others=others)
    assert not rccl and ctx.log == ["comm_destroy", "set_exchange"] and x["transport"] == "host"
    assert "failed on some rank" in gather


def test_compare_keeps_the_faster_of_the_two_and_drops_the_other():
    ctx, sw, x, gather, rccl = _choose("compare", _Sweep(), {"p2p": 5.0, "rccl": 25.0, "host": 90.0})
    assert sw.calls == ["init_rccl", "init_p2p"] and x["transport"] == "p2p" and rccl and ctx.log == []   # kept: the final gather's
    assert x["rccl_us"] == 25.0 and x["p2p_us"] == 5.0
    ctx, sw, x, gather, rccl = _choose("compare", _Sweep(), {"p2p": 40.0, "rccl": 25.0, "host": 90.0})
    assert x["transport"] == "rccl" and rccl and ctx.log == ["p2p_detach"] and x["exchange_us"] == 25.0


def test_the_slowest_ranks_probe_counts_and_a_failed_probe_is_reported():
    others = lambda v: [v * 3] if isinstance(v, float) else [v]
    ctx, sw, x, gather, rccl = _choose("auto", _Sweep(), {"p2p": 5.0, "rccl": 25.0, "host": 90.0}, others=others)
    assert x["p2p_us"] == 15.0
    ctx, sw, x, gather, rccl = _choose("p2p", _Sweep(), {"p2p": _Err("exchange timed out"), "rccl": 25.0, "host": 90.0})
    assert x["p2p_us"] is None and x["probe_errors"] == ["exchange timed out"] and x["transport"] == "p2p"


def test_theta_dealing_only_wants_a_gather():
    ctx, sw, x, gather, rccl = _choose("auto", _Sweep(), {"p2p": 5.0, "rccl": 25.0, "host": 90.0}, nshard=False)
    assert sw.calls == ["init_rccl"] and rccl and gather == "rccl-allgather" and x == {}


# ---- r06: `python3 bench.py --gpus N` launches its own ranks (bench.launch_ranks), no torchrun, no HIP call in the launcher ----
_CHILD = r'''
import json, os, sys, time
r = int(os.environ["RANK"])
d = os.environ["FAKE_DIR"]
with open(os.path.join(d, "env_%d.json" % r), "w") as fp:
    json.dump({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                               "TORCHELASTIC_RUN_ID", "BIOEN_BENCH_LAUNCHED")}, fp)
print("line of rank %d" % r)
sys.stdout.flush()
time.sleep(float(os.environ.get("FAKE_SLEEP_%d" % r, "0")))
sys.exit(int(os.environ.get("FAKE_EXIT_%d" % r, "0")))
'''


def _launch(tmp_path, monkeypatch, n, timeout=30.0, **env):
    import io
    import json
    child = tmp_path / "child.py"
    child.write_text(_CHILD)
    monkeypatch.setenv("FAKE_DIR", str(tmp_path))
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    out, err = io.StringIO(), io.StringIO()
    rc = bench.launch_ranks(n, [sys.executable, str(child)], timeout, out=out, err=err)
    envs = {}
    for r in range(n):
        f = tmp_path / ("env_%d.json" % r)
        if f.exists():
            envs[r] = json.loads(f.read_text())
    return rc, out.getvalue(), err.getvalue(), envs


def test_launcher_starts_n_ranks_with_the_rank_environment_and_relays_rank_zero(tmp_path, monkeypatch):
    rc, out, err, envs = _launch(tmp_path, monkeypatch, 4)
    assert rc == 0 and sorted(envs) == [0, 1, 2, 3]
    for r, e in envs.items():
        assert e["RANK"] == str(r) and e["LOCAL_RANK"] == str(r) and e["WORLD_SIZE"] == "4" and e["LOCAL_WORLD_SIZE"] == "4"
        assert e["MASTER_ADDR"] == "127.0.0.1" and e["BIOEN_BENCH_LAUNCHED"] == "1"
    assert len({e["MASTER_PORT"] for e in envs.values()}) == 1 and len({e["TORCHELASTIC_RUN_ID"] for e in envs.values()}) == 1
    assert out == "line of rank 0\n"                                  # stdout carries rank 0's line and nothing else
    assert all(("line of rank %d" % r) in err for r in (1, 2, 3))


def test_launcher_propagates_the_first_failure_and_stops_the_other_ranks(tmp_path, monkeypatch):
    import time
    t0 = time.time()
    rc, out, err, envs = _launch(tmp_path, monkeypatch, 3, FAKE_EXIT_1=7, FAKE_SLEEP_0=60, FAKE_SLEEP_2=60)
    assert rc == 7 and "rank 1 left with status 7" in err
    assert time.time() - t0 < 30.0                                    # ranks 0 and 2 were not waited for


def test_launcher_kills_the_ranks_at_its_time_bound(tmp_path, monkeypatch):
    rc, out, err, envs = _launch(tmp_path, monkeypatch, 2, timeout=1.0, FAKE_SLEEP_0=60, FAKE_SLEEP_1=60)
    assert rc == 124 and "did not finish" in err


_FAKE_PKG = r'''
import json, os, sys
def device_count():
    return int(os.environ.get("FAKE_NDEV", "1"))
if os.environ.get("BIOEN_BENCH_LAUNCHED") == "1" or "RANK" in os.environ:      # a rank of the bench: record, answer, leave
    r = int(os.environ.get("RANK", "0"))
    with open(os.path.join(os.environ["FAKE_DIR"], "rank_%d.json" % r), "w") as fp:
        json.dump({"world": os.environ.get("WORLD_SIZE"), "argv": sys.argv[1:]}, fp)
    if r == 0:
        print(json.dumps({"n_gpus": int(os.environ["WORLD_SIZE"])}))
    sys.exit(int(os.environ.get("FAKE_EXIT_%d" % r, "0")))
'''


def _bench_copy(tmp_path, args, **env):
    """bench.py itself, run as a program beside a FAKE bioen_amd (no GPU, no library): what does `--gpus N` do?"""
    import shutil
    import subprocess
    shutil.copy(os.path.join(ROOT, "bench.py"), tmp_path / "bench.py")
    pkg = tmp_path / "bioen_amd"
    pkg.mkdir(exist_ok=True)
    (pkg / "__init__.py").write_text(_FAKE_PKG)
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(FAKE_DIR=str(tmp_path), PYTHONPATH="")
    e.update({k: str(v) for k, v in env.items()})
    p = subprocess.run([sys.executable, str(tmp_path / "bench.py")] + args, env=e, capture_output=True, text=True, timeout=120)
    ranks = sorted(int(f.name[5:-5]) for f in tmp_path.glob("rank_*.json"))
    return p, ranks


def test_bench_with_gpus_n_and_no_launcher_becomes_the_launcher(tmp_path):
    import json
    p, ranks = _bench_copy(tmp_path, ["--gpus", "3", "--steps", "2"], FAKE_NDEV=4)
    assert p.returncode == 0, p.stderr
    assert ranks == [0, 1, 2]
    assert json.loads(p.stdout.strip().splitlines()[-1]) == {"n_gpus": 3}
    rec = json.loads((tmp_path / "rank_2.json").read_text())
    assert rec["world"] == "3" and rec["argv"] == ["--gpus", "3", "--steps", "2"]          # the ranks get the caller's arguments


def test_bench_refuses_fewer_devices_than_ranks_unless_told_to_share(tmp_path):
    p, ranks = _bench_copy(tmp_path, ["--gpus", "2"], FAKE_NDEV=1)
    assert p.returncode == 5 and ranks == [] and "only 1 device(s) visible" in p.stderr and p.stdout == ""
    p, ranks = _bench_copy(tmp_path, ["--gpus", "2", "--share-devices"], FAKE_NDEV=1)
    assert p.returncode == 0 and ranks == [0, 1]


def test_bench_passes_a_ranks_failure_on(tmp_path):
    p, ranks = _bench_copy(tmp_path, ["--gpus", "2"], FAKE_NDEV=2, FAKE_EXIT_1=4)
    assert p.returncode == 4


def test_bench_refuses_a_label_that_is_not_the_rank_count(tmp_path):
    p, ranks = _bench_copy(tmp_path, ["--gpus", "4"], FAKE_NDEV=8, WORLD_SIZE=2, RANK=0)
    assert p.returncode == 5 and "refusing to run under a wrong label" in p.stderr


def test_check_common_form_warns_when_one_rank_fell_back_alone():
    """a rank that takes the one-copy form by itself leaves the bit-identity across GPU counts -- the sharded
    drivers compare the form over the ranks (sweep.check_common_form)."""
    import warnings
    from bioen_amd import sweep

    class Ctx(object):
        def __init__(self, one):
            self.one = one

        def layout(self):
            return {"one_copy": self.one, "interleave": 1, "relayouts": 0}

    class Comm(object):
        world = 4

        def __init__(self, forms):
            self.forms = forms

        def allgather_object(self, obj):
            return list(self.forms)

    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert sweep.check_common_form(Ctx(0), Comm([0, 0, 0, 0])) == {"one_copy": [0, 0, 0, 0], "common": True}
        assert sweep.check_common_form(Ctx(1), Comm([1, 1, 1, 1]))["common"]
        assert sweep.check_common_form(Ctx(1), None) == {"one_copy": [1], "common": True}
    with pytest.warns(RuntimeWarning, match="ranks \\[2\\]"):
        out = sweep.check_common_form(Ctx(0), Comm([0, 0, 1, 0]))
    assert out == {"one_copy": [0, 0, 1, 0], "common": False}


def test_wall_budget_drops_lowest_priority_records_and_keeps_the_reserve():
    """r06: the side records of bench.py run under a wall budget -- a droppable record that no longer
    fits (its estimate plus what the never-dropped ones still need) is left out and named; the never-dropped ones run."""
    now = [0.0]
    b = bench.WallBudget(100.0, reserve=30.0, clock=lambda: now[0])
    assert b.take("first", 50.0)                     # 100 - 30 >= 50
    with b.timed("first"):
        now[0] += 55.0
    assert not b.take("second", 20.0)                # 45 left, 30 reserved: 15 < 20
    sk = b.skipped(20.0)
    assert sk["skipped"] == "budget" and sk["budget_left_s"] == 45.0 and sk["reserved_s"] == 30.0
    assert b.take("third", 10.0)                     # a cheaper one still fits
    with b.timed("third"):
        now[0] += 8.0
    b.reserve = 0.0                                  # the never-dropped record runs now, whatever is left
    with b.timed("never_dropped"):
        now[0] += 50.0
    rep = b.report()
    assert rep["skipped_for_budget"] == ["second"] and rep["used_s"] == 113.0
    assert rep["seconds"] == {"first": 55.0, "third": 8.0, "never_dropped": 50.0}
    assert not b.take("late", 1.0) and b.left() == -13.0
