set -e
mkdir -p gpurun_out/r06_fwd_ab
for rep in 1 2; do
for v in default fwddiag1 fwddiag2 fwddiag3; do
  if [ $v = default ]; then unset BIOEN_HIP_LIBRARY; else export BIOEN_HIP_LIBRARY=$PWD/build/libbioen_$v.so; fi
  timeout -k 10 200 python3 tools/pass_probe.py 1024 1000000 30 >> gpurun_out/r06_fwd_ab/$v.json
done
unset BIOEN_HIP_LIBRARY
BIOEN_HIP_STRIP_FOLD=0 timeout -k 10 200 python3 tools/pass_probe.py 1024 1000000 30 >> gpurun_out/r06_fwd_ab/default_nofold.json
done
cat gpurun_out/r06_fwd_ab/*.json
