for lib in "" build/libbioen_precentered.so; do
  echo "== lib: ${lib:-default (raw copies, in-kernel centring)}"
  BIOEN_HIP_LIBRARY=$lib REPS=30 python tools/strip_probe.py | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print({k:(round(v['xy_ms'],4), round(v['bt_ms'],4)) for k,v in d['K'].items()})"
  BIOEN_HIP_LIBRARY=$lib SIZES=1024:1000000,256:100000 python tools/small_timeline.py
done
