cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c10; mkdir -p $O; rm -f $O/*
python -m pytest tests/test_hip_api.py -m gpu -q -x -k "uploads or find_optimum" > $O/tests.log 2>&1; tail -n 3 $O/tests.log
( time python bench.py ) > $O/bench_N1.json 2> $O/bench_N1.err; tail -n 4 $O/bench_N1.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --no-cpu-baseline > $O/bench_2r.json 2> $O/bench_2r.err; tail -n 3 $O/bench_2r.err
python - <<'PY'
import json
for f in ("gpurun_out/c10/bench_N1.json","gpurun_out/c10/bench_2r.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f,"ERR",e); continue
    print(f, d["value"], d["ms_per_step"], d["iterations_per_sweep"], d["roofline"]["frac"] if d.get("roofline") else None)
    c=d["config"]; print("  rccl_ranks",c.get("rccl_ranks"),"rccl_error",c.get("rccl_error"),"transport",c.get("exchange_transport"),"final_gather",c.get("final_gather"))
    if d.get("api_end_to_end"):
        a=d["api_end_to_end"]
        for k in ("headline","configs1"):
            if k in a: print("  api",k,{kk:(vv if not isinstance(vv,dict) else {x:y for x,y in vv.items() if x not in ("fmin","per_theta_s")}) for kk,vv in a[k].items() if kk!="workload"})
        print("  api err", a.get("error"))
    cb=d.get("cpu_baseline") or {}
    print("  full_size mid:", (cb.get("full_size") or {}).get("mid_theta"))
    f_=d.get("forces") or {}
    print("  forces:", f_.get("value"), f_.get("ms_per_step"), f_.get("thetas_failed"), f_.get("failed_thetas_vs_reference"))
PY
