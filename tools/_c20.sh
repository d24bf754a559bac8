OUT=gpurun_out/c20
mkdir -p $OUT
python -m pytest tests/test_hip_fullsize.py tests/test_hip_api.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -n 2 $OUT/tests.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 1 --warmup 1 > $OUT/bench_2ranks_one_gpu.json 2> $OUT/bench_2ranks.err; echo "2 ranks rc $?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29522 bench.py --gpus 4 --steps 1 --warmup 1 > $OUT/bench_4ranks_one_gpu.json 2> $OUT/bench_4ranks.err; echo "4 ranks rc $?"
tail -c 600 $OUT/bench_2ranks_one_gpu.json; echo
FIRST=400 SEEDS=800 timeout -k 10 600 python tools/fuzz_canon.py > $OUT/fuzz_canon.log 2>&1; echo "fuzz_canon rc $?"; tail -n 1 $OUT/fuzz_canon.log
FIRST=100 SEEDS=200 timeout -k 10 400 python tools/fuzz_batch.py > $OUT/fuzz_batch.log 2>&1; echo "fuzz_batch rc $?"; tail -n 1 $OUT/fuzz_batch.log
FIRST=100 SEEDS=200 MIN_DIM=4 timeout -k 10 400 python tools/fuzz_parity.py > $OUT/fuzz_parity.log 2>&1; echo "fuzz_parity rc $?"; tail -n 2 $OUT/fuzz_parity.log
FIRST=100 SEEDS=100 timeout -k 10 400 python tools/fuzz_api.py > $OUT/fuzz_api.log 2>&1; echo "fuzz_api rc $?"; tail -n 1 $OUT/fuzz_api.log
FIRST=100 SEEDS=100 timeout -k 10 300 python tools/fuzz_gsl.py > $OUT/fuzz_gsl.log 2>&1; echo "fuzz_gsl rc $?"; tail -n 1 $OUT/fuzz_gsl.log
FIRST=100 SEEDS=150 timeout -k 10 300 python tools/fuzz_last_average.py > $OUT/fuzz_last_average.log 2>&1; echo "fuzz_last_average rc $?"; tail -n 1 $OUT/fuzz_last_average.log
FIRST=100 SEEDS=100 timeout -k 10 300 python tools/fuzz_context.py > $OUT/fuzz_context.log 2>&1; echo "fuzz_context rc $?"; tail -n 1 $OUT/fuzz_context.log
