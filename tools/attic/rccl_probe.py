"""Development aid: single-rank RCCL communicator through the C ABI (dlopen path, all-gather)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
rng = np.random.default_rng(0)
ctx = bioen_amd.Context(rng.normal(size=(4, 300)), np.zeros(4))
uid = ctx.comm_unique_id()
print("unique id bytes", len(uid))
ctx.comm_init(uid, 0, 1)
x = rng.normal(size=1000)
out = ctx.comm_allgather(x, 1)
assert out.shape == (1, 1000) and np.array_equal(out[0], x)
print("rccl single-rank allgather ok")
ctx.close()
