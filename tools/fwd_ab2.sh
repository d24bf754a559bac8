set -e
mkdir -p gpurun_out/r06_fwd_ab2
FLAGS="--no-cpu-baseline --no-forces --no-api --no-deer --no-ala5 --no-matched --no-pmc --no-storage-experiment --no-one-copy"
timeout -k 10 200 python3 tools/pass_probe.py 1024 1000000 30 > gpurun_out/r06_fwd_ab2/probe.json
timeout -k 10 300 python3 bench.py $FLAGS --steps 2 > gpurun_out/r06_fwd_ab2/bench_default.json
BIOEN_HIP_DEVICE_LS=1 timeout -k 10 300 python3 bench.py $FLAGS --steps 2 > gpurun_out/r06_fwd_ab2/bench_devls.json
timeout -k 10 200 python3 tools/pass_probe.py 1024 1000000 30 >> gpurun_out/r06_fwd_ab2/probe.json
python3 - <<'PY'
import json
for f in ("bench_default","bench_devls"):
    d=json.loads(open("gpurun_out/r06_fwd_ab2/%s.json"%f).read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(f, d["ms_per_step"], d["iterations_per_sweep"], {n:(round(v["avg_ms"],4), round(v["avg_batch_width"],2), v["launches"]) for n,v in k.items()}, d["roofline"]["read_ceiling"]["GB/s"])
PY
cat gpurun_out/r06_fwd_ab2/probe.json
