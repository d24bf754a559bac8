set -e
mkdir -p gpurun_out/r06_onecopy2
rm -f gpurun_out/r06_onecopy2/*.json
for shape in "256 100000" "205 500000" "28 50001" "64 20000" "512 1000000" "1024 125000" "600 200000" "2100 100000"; do
  for oc in 0 1; do
    BIOEN_HIP_ONE_COPY=$oc KS=1,4,8 timeout -k 10 200 python3 tools/pass_probe.py $shape 40 >> gpurun_out/r06_onecopy2/oc$oc.json
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_onecopy2/*.json')):
    for ln in open(f):
        d=json.loads(ln)
        print(f.split('/')[-1][:-5].ljust(6), ("%dx%d"%(d['M'],d['N'])).ljust(14), ' '.join('K%s fwd %s adj %s |'%(k,'/'.join('%.4f'%x['fwd_ms'] for x in v),'/'.join('%.4f'%x['adj_ms'] for x in v)) for k,v in d['K'].items()))
PY
