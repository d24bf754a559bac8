set -e
mkdir -p gpurun_out/r06_r04_ab
rm -f gpurun_out/r06_r04_ab/*.json
FLAGS="--no-cpu-baseline --no-forces --no-api --no-deer --no-ala5 --no-matched --no-pmc --no-storage-experiment --no-one-copy"
for rep in 1 2; do
  (cd build/r04tree && timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-forces --no-deer --no-ala5 --no-matched --no-pmc --no-storage-experiment --steps 2) > gpurun_out/r06_r04_ab/r04_$rep.json
  BIOEN_HIP_STRIP_INTERLEAVE=0 timeout -k 10 300 python3 bench.py $FLAGS --steps 2 > gpurun_out/r06_r04_ab/r06_ilv0_$rep.json
  timeout -k 10 300 python3 bench.py $FLAGS --steps 2 > gpurun_out/r06_r04_ab/r06_ilv1_$rep.json
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_r04_ab/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d["roofline"]["kernels"]
    print(f.split('/')[-1].ljust(18), "sweep %.1f ms"%d["ms_per_step"], d["iterations_per_sweep"], {n:(round(v["avg_ms"],4), round(v["avg_batch_width"],2), v["launches"]) for n,v in k.items()}, "read ceiling %.0f"%d["roofline"]["read_ceiling"]["GB/s"])
PY
