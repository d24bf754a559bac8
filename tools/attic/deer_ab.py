"""Development aid: the DEER record of bench.py (a K = 1 series: every theta runs alone, refits in between) under the
engine / shadow switches, one process, interleaved."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd, bench

VARIANTS = {
    "device": {},
    "device-shadows2": {"BIOEN_HIP_SHADOWS": "2", "BIOEN_HIP_SHADOW_RATE": "0"},
    "device-shadows4": {"BIOEN_HIP_SHADOWS": "4", "BIOEN_HIP_SHADOW_RATE": "0"},
    "host": {"BIOEN_HIP_DEVICE_LS": "0"},
    "host-nospec": {"BIOEN_HIP_DEVICE_LS": "0", "BIOEN_HIP_SPECULATE": "0"},
}
KEYS = sorted({k for v in VARIANTS.values() for k in v})
for rep in range(int(os.environ.get("REPS", "2"))):
    for name, env in VARIANTS.items():
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        r = bench.deer_record(bioen_amd, bench.SEED)
        print("%-16s %.3f s, %d iterations, %d evaluations, fmin %s" % (name, r["seconds"], r["iterations"], r["evaluations"], ["%.10g" % f for f in r["fmin"]]))
        sys.stdout.flush()
