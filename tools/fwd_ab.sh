set -e
mkdir -p gpurun_out/r06_ilv2
timeout -k 10 300 python3 tools/strip_probe.py > gpurun_out/r06_ilv2/forces.json
FORCES_M=1024 timeout -k 10 300 python3 tools/strip_probe.py > gpurun_out/r06_ilv2/forces1024.json
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_ilv2/forces*.json')):
    for ln in open(f):
        d=json.loads(ln)
        print(f.split('/')[-1][:-5].ljust(16), ' '.join('K%s xy %.3f bt %.3f eq %s |'%(k,v['xy_ms'],v['bt_ms'],v['bitwise_equal_to_single']) for k,v in d['K'].items()))
PY
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -8
