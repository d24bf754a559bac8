"""TEST INFRASTRUCTURE ONLY -- how many CPUs this process may really use
(affinity mask capped by the cgroup CPU quota), so that OpenMP teams of the CPU
checkers are not oversubscribed on a GPU box that shows 256 cores but grants 16."""
import math
import os


def usable_cpus():
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fp:
                txt = fp.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(math.ceil(float(txt[0]) / float(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                        n = min(n, max(1, int(math.ceil(q / float(fp.read())))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def default_omp_threads():
    """Called before a checker library is loaded: give OpenMP a sane team size unless the
    user already chose one."""
    if "OMP_NUM_THREADS" not in os.environ:
        os.environ["OMP_NUM_THREADS"] = str(usable_cpus())
    return int(os.environ["OMP_NUM_THREADS"])
