"""theta-series (confidence-parameter sweep) sharded over the GPUs of one node.

The reference runs the series serially (``bioen/analyze/procedure.py:62-83``:
``for theta in options.thetas``).  Every theta is an independent L-BFGS problem
over the same read-only (yTilde, YTilde, G), so the series shards with no
communication inside the loop: rank r solves its thetas against its own
HBM-resident copy of yTilde, and ONE all-gather at the end ships
(theta, fmin, chi2, S, iterations, evaluations, status, seconds, w[N]) per theta
to every rank -- over RCCL/xGMI when the context has an RCCL communicator,
else over the control-plane communicator (which is what the CPU tests use).

Control plane (rendezvous, barrier, max-over-ranks): ``SocketComm`` (pure-Python
TCP, no torch -- the GPU processes stay torch-free) or ``TorchComm`` (any
``torch.distributed`` process group, e.g. gloo in the tests).
"""
import hmac
import json
import os
import secrets
import socket
import stat
import struct
import time

import numpy as np

from ._lib import BioenHipError

HEADER = 8   # doubles in front of w[N]: theta, fmin, chi2, kl, iterations, evaluations, code, seconds


# ------------------------------------------------------------------------------------
# sharding
# ------------------------------------------------------------------------------------
def shard_thetas(thetas, rank, world):
    """Indices (into `thetas`) that `rank` solves.

    Small theta = weak prior = more L-BFGS iterations, so the thetas are dealt
    round-robin in ascending order: the expensive ones land on different ranks."""
    order = np.argsort(np.asarray(thetas, dtype=np.float64), kind="stable")
    return [int(i) for k, i in enumerate(order) if k % world == rank]


# ------------------------------------------------------------------------------------
# communicators (control plane)
# ------------------------------------------------------------------------------------
class SingleComm(object):
    rank, world = 0, 1

    def allgather_array(self, a):
        return np.asarray(a, dtype=np.float64).reshape(1, -1).copy()

    def allgather_object(self, obj):
        return [obj]

    def barrier(self):
        pass

    def max(self, x):
        return float(x)

    def close(self):
        pass


class TorchComm(object):
    """Adapter over an initialised ``torch.distributed`` process group (CPU tests: gloo)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self._group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def allgather_array(self, a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).reshape(-1))
        out = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(out, t, group=self._group)
        return np.stack([o.numpy() for o in out])

    def allgather_object(self, obj):
        out = [None] * self.world
        self._dist.all_gather_object(out, obj, group=self._group)
        return out

    def barrier(self):
        self._dist.barrier(group=self._group)

    def max(self, x):
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self._group)
        return float(t[0])

    def close(self):
        pass


class ThreadComm(object):
    """`world` ranks as THREADS of one process (``ThreadComm.create(world)`` -> one object per rank): the control plane and
    the host-staged stage exchange of structure-sharded contexts that all live on one GPU of one process.  What it is for:
    a GPU box admits a handful of processes per card, threads it does not count -- eight ranks, the decomposition of a
    full node, run here under test (tests/test_hip_nshard.py).  Same interface as SocketComm."""

    class _Shared(object):
        def __init__(self, world, timeout):
            import threading
            self.world, self.timeout = world, timeout
            self.slots = [None] * world
            self.barrier = threading.Barrier(world)

    def __init__(self, shared, rank):
        self._s, self.rank, self.world = shared, rank, shared.world

    @classmethod
    def create(cls, world, timeout=120.0):
        shared = cls._Shared(world, timeout)
        return [cls(shared, r) for r in range(world)]

    def allgather_object(self, obj):
        s = self._s
        s.slots[self.rank] = obj
        s.barrier.wait(s.timeout)              # everybody has written
        out = list(s.slots)
        s.barrier.wait(s.timeout)              # everybody has read: the slots may be rewritten
        return out

    def allgather_array(self, a):
        return np.stack(self.allgather_object(np.array(a, dtype=np.float64, copy=True)))

    def barrier(self):
        self._s.barrier.wait(self._s.timeout)

    def max(self, x):
        return max(self.allgather_object(float(x)))

    def close(self):
        pass


def _enc(obj):
    """Small control objects only (None, bool, numbers, str, bytes, lists / tuples of those): JSON, no pickle."""
    def conv(o):
        if isinstance(o, (bytes, bytearray)):
            return {"__bytes__": bytes(o).hex()}
        if isinstance(o, (list, tuple)):
            return [conv(x) for x in o]
        if isinstance(o, (np.floating, np.integer, np.bool_)):
            return o.item()
        if o is None or isinstance(o, (bool, int, float, str)):
            return o
        raise TypeError("SocketComm.allgather_object: unsupported type %r" % type(o))
    return json.dumps(conv(obj)).encode("utf-8")


def _dec(payload):
    def conv(o):
        if isinstance(o, dict):
            if set(o) != {"__bytes__"}:
                raise ValueError("SocketComm: malformed control message")
            return bytes.fromhex(o["__bytes__"])
        if isinstance(o, list):
            return [conv(x) for x in o]
        return o
    return conv(json.loads(payload.decode("utf-8")))


class SocketComm(object):
    """Star-topology TCP communicator for the ranks of ONE node, driven by the
    torchrun environment (RANK, WORLD_SIZE, MASTER_PORT).

    Rank 0 listens on an ephemeral LOOPBACK port and publishes port + a random 32-byte token in a
    rendezvous file inside a directory only this user can enter (mode 0700, ownership checked, file
    created with O_EXCL | O_NOFOLLOW).  A peer is accepted only after presenting the token; messages
    are length-prefixed raw float64 buffers or JSON for the small control objects -- nothing that
    arrives from the network is ever unpickled."""

    MAGIC = b"BIOENAMD"
    MAX_MSG = 1 << 34

    @staticmethod
    def _private_dir():
        d = os.path.join("/tmp", "bioen_amd_%d" % os.getuid())
        try:
            os.mkdir(d, 0o700)
        except FileExistsError:
            pass
        st = os.lstat(d)
        if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise RuntimeError("SocketComm: %s is not a private directory of this user" % d)
        return d

    def __init__(self, rank=None, world=None, timeout=300.0):
        self.rank = int(os.environ.get("RANK", 0) if rank is None else rank)
        self.world = int(os.environ.get("WORLD_SIZE", 1) if world is None else world)
        self._peers = []
        self._sock = None
        if self.world == 1:
            return
        port = os.environ.get("MASTER_PORT", "29500")
        run_id = os.environ.get("TORCHELASTIC_RUN_ID", "none")
        safe = "".join(ch if ch.isalnum() else "_" for ch in "%s_%s" % (port, run_id))
        path = os.path.join(self._private_dir(), "rdzv_" + safe)
        deadline = time.time() + timeout
        if self.rank == 0:
            token = secrets.token_bytes(32)
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(self.world)
            try:
                os.unlink(path)                   # a stale file of an earlier run of ours (the directory is private)
            except FileNotFoundError:
                pass
            fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
            with os.fdopen(fd, "w") as fp:
                fp.write("%d %s\n" % (srv.getsockname()[1], token.hex()))
            self._path = path
            peers = {}
            srv.settimeout(timeout)
            while len(peers) < self.world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                try:
                    hello = self._recv_exact(conn, len(self.MAGIC) + 4 + 32)
                except (RuntimeError, OSError):
                    conn.close()
                    continue
                r = struct.unpack("<i", hello[len(self.MAGIC):len(self.MAGIC) + 4])[0]
                if (hello[:len(self.MAGIC)] != self.MAGIC or not hmac.compare_digest(hello[len(self.MAGIC) + 4:], token)
                        or not 0 < r < self.world or r in peers):
                    conn.close()
                    continue
                conn.sendall(self.MAGIC)
                peers[r] = conn
            srv.close()
            self._peers = [peers[r] for r in range(1, self.world)]
        else:
            while True:
                try:
                    fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
                    with os.fdopen(fd) as fp:
                        if os.fstat(fp.fileno()).st_uid != os.getuid():
                            raise ValueError("foreign rendezvous file")
                        fields = fp.read().split()
                    p, token = int(fields[0]), bytes.fromhex(fields[1])
                    s = socket.create_connection(("127.0.0.1", p), timeout=5.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.sendall(self.MAGIC + struct.pack("<i", self.rank) + token)
                    s.settimeout(timeout)
                    if self._recv_exact(s, len(self.MAGIC)) == self.MAGIC:
                        self._sock = s
                        break
                    s.close()
                except (OSError, ValueError, IndexError, RuntimeError):
                    pass
                if time.time() > deadline:
                    raise RuntimeError("SocketComm: rendezvous with rank 0 timed out (%s)" % path)
                time.sleep(0.05)

    @staticmethod
    def _recv_exact(s, n):
        buf = bytearray()
        while len(buf) < n:
            chunk = s.recv(min(1 << 20, n - len(buf)))
            if not chunk:
                raise RuntimeError("SocketComm: peer closed the connection")
            buf += chunk
        return bytes(buf)

    def _send_msg(self, s, payload):
        s.sendall(struct.pack("<q", len(payload)) + payload)

    def _recv_msg(self, s):
        n = struct.unpack("<q", self._recv_exact(s, 8))[0]
        if n < 0 or n > self.MAX_MSG:
            raise RuntimeError("SocketComm: bad message length")
        return self._recv_exact(s, n)

    def _allgather_bytes(self, payload):
        """-> [payload of rank 0, ..., payload of rank world-1]; framing: count, lengths, concatenated bytes"""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [self._recv_msg(p) for p in self._peers]
            blob = struct.pack("<q", len(parts)) + b"".join(struct.pack("<q", len(x)) for x in parts) + b"".join(parts)
            for p in self._peers:
                self._send_msg(p, blob)
            return parts
        self._send_msg(self._sock, payload)
        blob = self._recv_msg(self._sock)
        cnt = struct.unpack_from("<q", blob, 0)[0]
        if cnt != self.world:
            raise RuntimeError("SocketComm: malformed gather message")
        lens = struct.unpack_from("<%dq" % cnt, blob, 8)
        off = 8 + 8 * cnt
        if any(x < 0 for x in lens) or off + sum(lens) != len(blob):
            raise RuntimeError("SocketComm: malformed gather message")
        parts = []
        for x in lens:
            parts.append(blob[off:off + x])
            off += x
        return parts

    def allgather_array(self, a):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float64)).reshape(-1)
        parts = self._allgather_bytes(a.tobytes())
        return np.stack([np.frombuffer(p, dtype=np.float64) for p in parts])

    def allgather_object(self, obj):
        return [_dec(p) for p in self._allgather_bytes(_enc(obj))]

    def barrier(self):
        self._allgather_bytes(b"")

    def max(self, x):
        return float(max(self.allgather_object(float(x))))

    def close(self):
        for p in self._peers:
            p.close()
        if self._sock is not None:
            self._sock.close()
        if self.rank == 0 and self.world > 1:
            try:
                os.remove(self._path)
            except OSError:
                pass
        self._peers, self._sock = [], None


def init_rccl(ctx, comm):
    """Create the RCCL communicator of `ctx` (one rank per GPU): rank 0 draws the
    ncclUniqueId, the control plane ships its 128 bytes."""
    if comm.world == 1:
        return False
    uid, err = None, None
    if comm.rank == 0:
        try:
            uid = ctx.comm_unique_id()
        except BioenHipError as e:        # e.g. librccl not loadable: every rank must learn it, none may wait
            err = str(e)
    uid, err = comm.allgather_object((uid, err))[0]
    if uid is None:
        raise BioenHipError("RCCL unavailable on rank 0: %s" % err)
    ctx.comm_init(uid, comm.rank, comm.world)
    return True


def init_p2p(ctx, comm, selftest=400):
    """Attach the peer-to-peer stage exchange of a structure-sharded `ctx` (one rank per process): every rank exports the
    hipIpc handle of its mailbox, the control plane all-gathers the 64-byte handles, every rank maps its peers.  Ranks
    agree on the outcome: if any rank cannot export, map or pass the self-test, ALL detach and the function returns
    False (the caller falls back to RCCL / the host-staged path)."""
    if comm.world == 1:
        return False
    handle, err = None, None
    try:
        handle = ctx.p2p_export()
    except BioenHipError as e:
        err = str(e)
    got = comm.allgather_object((handle, err))
    ok = all(h is not None for h, _ in got)
    if ok:
        try:
            ctx.p2p_attach([h for h, _ in got])
        except BioenHipError:
            ok = False
    ok = all(comm.allgather_object(ok))
    if ok and selftest:
        try:
            ok = ctx.exchange_selftest(selftest) == 0
        except BioenHipError:
            ok = False
        ok = all(comm.allgather_object(ok))
    if not ok:
        try:
            ctx.p2p_detach()
        except BioenHipError:
            pass
    return ok


# ------------------------------------------------------------------------------------
# the sweep
# ------------------------------------------------------------------------------------
def _pack(theta, w, info, n):
    rec = np.zeros(HEADER + n)
    rec[:HEADER] = (theta, info.fmin, info.chi2, info.kl, info.iterations, info.evaluations,
                    info.lbfgs_code, info.seconds)
    rec[HEADER:] = w
    return rec


def theta_sweep(ctx, thetas, solve, comm=None, rccl=False, n=None, presolved=None):
    """Solve every theta of the series, sharded over comm.world ranks, and gather.

    ctx    : bioen_amd.Context (may be None when `solve` does not need one and rccl is False)
    solve  : callable(theta) -> (w[n], info) with info.{fmin,chi2,kl,iterations,evaluations,
             lbfgs_code,seconds}; the product passes a closure over ctx.opt_lbfgs_logw /
             ctx.opt_lbfgs_forces, the CPU tests inject their checker.
    rccl   : gather through ctx.comm_allgather (RCCL over xGMI) instead of `comm`.
    presolved : optional {index into thetas: (w, info)} for this rank's thetas (a rank that
             solved its shard as one lock-step batch); `solve` is then not called.
    Returns a list (in the order of `thetas`) of dicts, identical on every rank.
    """
    comm = comm or SingleComm()
    thetas = [float(t) for t in thetas]
    n = ctx.n if n is None else n
    mine = shard_thetas(thetas, comm.rank, comm.world)
    if comm.world == 1:
        # one rank: nothing to gather -- the records are built around the solver's own arrays (packing 2 x 8 MB per theta
        # into a gather buffer and copying them out again cost 15-20 ms of the headline sweep's 1.1 s)
        out = [None] * len(thetas)
        for idx in mine:
            w, info = presolved[idx] if presolved is not None else solve(thetas[idx])
            out[idx] = {"theta": thetas[idx], "fmin": info.fmin, "chi2": info.chi2, "S": -info.kl,
                        "iterations": int(info.iterations), "evaluations": int(info.evaluations), "code": int(info.lbfgs_code),
                        "seconds": info.seconds, "rank": 0, "w": np.asarray(w, dtype=np.float64).reshape(-1)}
        return out
    per_rank = -(-len(thetas) // comm.world)          # ceil: fixed-size gather payload
    buf = np.zeros((per_rank, HEADER + n))
    buf[:, 0] = np.nan                                 # unused slots are marked by theta = NaN
    for slot, idx in enumerate(mine):
        w, info = presolved[idx] if presolved is not None else solve(thetas[idx])
        buf[slot] = _pack(thetas[idx], np.asarray(w, dtype=np.float64).reshape(-1), info, n)

    if comm.world == 1:
        gathered = buf.reshape(1, -1)
    elif rccl:
        gathered = ctx.comm_allgather(buf.reshape(-1), comm.world)
    else:
        gathered = comm.allgather_array(buf.reshape(-1))
    gathered = gathered.reshape(comm.world, per_rank, HEADER + n)

    out = [None] * len(thetas)
    for r in range(comm.world):
        for slot, idx in enumerate(shard_thetas(thetas, r, comm.world)):
            rec = gathered[r, slot]
            out[idx] = {"theta": rec[0], "fmin": rec[1], "chi2": rec[2], "S": -rec[3],
                        "iterations": int(rec[4]), "evaluations": int(rec[5]), "code": int(rec[6]),
                        "seconds": rec[7], "rank": r, "w": rec[HEADER:].copy()}
    return out


def gather_results(ctx, results, comm, rccl=True):
    """ONE all-gather of the final (theta, fmin, chi^2, S, iterations, evaluations, status, seconds, weights[n]) of every theta
    of `results` (a sweep's return value) over all ranks -- through RCCL (``ctx.comm_allgather``: ncclAllGather over xGMI)
    when `rccl`, else through the control plane.  On a structure-sharded run every rank already holds every result (the
    library gathers the vectors): the gather is then the cross-rank CONSISTENCY check -- every rank's record of every theta
    must be the same bytes.  -> {"seconds", "bytes_per_rank", "consistent", "ranks", "via"}"""
    import time
    n = results[0]["w"].size
    buf = np.empty((len(results), HEADER + n))
    for k, r in enumerate(results):
        buf[k, :HEADER] = (r["theta"], r["fmin"], r["chi2"], -r["S"], r["iterations"], r["evaluations"], r["code"], 0.0)
        buf[k, HEADER:] = r["w"]
    flat = buf.reshape(-1)
    t0 = time.perf_counter()
    if rccl:
        got = ctx.comm_allgather(flat, comm.world)
    else:
        got = comm.allgather_array(flat)
    dt = time.perf_counter() - t0
    got = np.asarray(got).reshape(comm.world, -1)
    same = all(np.array_equal(got[r].view(np.uint64), flat.view(np.uint64)) for r in range(comm.world))
    return {"seconds": dt, "bytes_per_rank": int(flat.nbytes), "consistent": bool(same), "ranks": int(comm.world),
            "via": "rccl" if rccl else "tcp"}


def check_common_form(ctx, comm):
    """A structure-sharded context promises the SAME bits on 1, 2, 4 and 8 ranks -- as long as every rank runs the same
    kernels.  A rank that could not allocate its second strip copy takes the one-copy form by itself
    (bioen_hip_ctx_layout: one_copy), whose adjoint rounds differently in the last bits: compare the form over the ranks
    and say so.  -> {"one_copy": [per rank], "common": bool}"""
    import warnings
    forms = comm.allgather_object(int(ctx.layout()["one_copy"])) if comm is not None and comm.world > 1 \
        else [int(ctx.layout()["one_copy"])]
    common = len(set(forms)) == 1
    if not common:
        warnings.warn("bioen_amd: ranks %s run the log-weights adjoint on ONE strip copy, the others on two: results agree to "
                      "rounding, not bit for bit, with runs on another number of GPUs (BIOEN_HIP_ONE_COPY=1 on every rank "
                      "gives one common form)" % [r for r, f in enumerate(forms) if f], RuntimeWarning)
    return {"one_copy": forms, "common": common}


def sweep_log_weights(ctx, thetas, G, g_init, lbfgs_params, comm=None, rccl=False, verbose=False, max_batch=8):
    """Cold-started log-weights series (every theta starts from g_init, as
    procedure.py:46,66 does for generic data).  The thetas of a rank run as ONE lock-step
    batch (ctx.opt_lbfgs_logw_batch): up to `max_batch` of them share every pass over yTilde."""
    comm = comm or SingleComm()
    thetas = [float(t) for t in thetas]
    mine = shard_thetas(thetas, comm.rank, comm.world)
    solved = {}
    if mine:
        _, w, infos = ctx.opt_lbfgs_logw_batch([thetas[i] for i in mine], g_init, G, lbfgs_params,
                                               max_batch=max_batch, verbose=verbose)
        for k, i in enumerate(mine):
            solved[i] = (w[k], infos[k])
    return theta_sweep(ctx, thetas, None, comm=comm, rccl=rccl, presolved=solved)


def sweep_log_weights_sharded(ctx, thetas, G, g_init, lbfgs_params, verbose=False, max_batch=8, comm=None):
    """The same series on a STRUCTURE-sharded context (``Context(..., rank, world)``): every rank
    keeps a column block of yTilde and all ranks work on every theta together -- each matrix
    pass, each N-vector kernel and each L-BFGS vector lives 1/world per GPU, and the reductions
    over structures are completed by one in-place all-gather per stage (RCCL over xGMI).  All
    ranks return the same list (results are gathered inside the library).  `comm` (a control-plane communicator, the
    same on every rank): the ranks' strip-copy forms are compared afterwards (check_common_form: a RuntimeWarning when one
    rank fell back to the one-copy form alone)."""
    thetas = [float(t) for t in thetas]
    _, w, infos = ctx.opt_lbfgs_logw_batch(thetas, g_init, G, lbfgs_params, max_batch=max_batch, verbose=verbose)
    if comm is not None:
        check_common_form(ctx, comm)          # (a control-plane collective: every rank passes the same `comm` or none)
    return [{"theta": th, "fmin": i.fmin, "chi2": i.chi2, "S": -i.kl, "iterations": i.iterations,
             "evaluations": i.evaluations, "code": i.lbfgs_code, "seconds": i.seconds, "rank": -1, "w": w[k]}
            for k, (th, i) in enumerate(zip(thetas, infos))]


def sweep_forces(ctx, thetas, w0, forces_init, lbfgs_params, comm=None, rccl=False, verbose=False, max_batch=8):
    """Cold-started forces series (the ala5 notebook's protocol); a rank's thetas run as one
    lock-step batch sharing the matrix passes of every evaluation (two for M <= 1024, else four)."""
    comm = comm or SingleComm()
    thetas = [float(t) for t in thetas]
    mine = shard_thetas(thetas, comm.rank, comm.world)
    solved = {}
    if mine:
        _, w, infos = ctx.opt_lbfgs_forces_batch([thetas[i] for i in mine], forces_init, w0, lbfgs_params,
                                                 max_batch=max_batch, verbose=verbose)
        for k, i in enumerate(mine):
            solved[i] = (w[k], infos[k])
    return theta_sweep(ctx, thetas, None, comm=comm, rccl=rccl, presolved=solved)


def sweep_forces_sharded(ctx, thetas, w0, forces_init, lbfgs_params, verbose=False, max_batch=8):
    """Forces series on a STRUCTURE-sharded context (BASELINE config 5's decomposition): every rank
    keeps a column block of yTilde and of w0, all thetas stay batched on every rank, and an
    evaluation needs two small all-gathers (the ranks' shares of ybar with their softmax totals, and
    of the gradient); the M force variables and the L-BFGS state are replicated and, fed with
    identical numbers, take identical decisions on every rank.  (M <= 1024: the two strip passes; beyond: the
    four passes over row panels -- two all-gathers per evaluation either way.)"""
    thetas = [float(t) for t in thetas]
    _, w, infos = ctx.opt_lbfgs_forces_batch(thetas, forces_init, w0, lbfgs_params, max_batch=max_batch, verbose=verbose)
    return [{"theta": th, "fmin": i.fmin, "chi2": i.chi2, "S": -i.kl, "iterations": i.iterations,
             "evaluations": i.evaluations, "code": i.lbfgs_code, "seconds": i.seconds, "rank": -1, "w": w[k]}
            for k, (th, i) in enumerate(zip(thetas, infos))]
