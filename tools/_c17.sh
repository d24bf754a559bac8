OUT=gpurun_out/c17
mkdir -p $OUT
SEEDS=60 timeout -k 10 500 python tools/fuzz_canon.py > $OUT/fuzz_canon.log 2>&1; echo "fuzz_canon rc $?"; tail -n 4 $OUT/fuzz_canon.log
SEEDS=40 timeout -k 10 240 python tools/fuzz_batch.py > $OUT/fuzz_batch.log 2>&1; echo "fuzz_batch rc $?"; tail -n 2 $OUT/fuzz_batch.log
SEEDS=60 timeout -k 10 240 python tools/fuzz_parity.py > $OUT/fuzz_parity.log 2>&1; echo "fuzz_parity rc $?"; tail -n 2 $OUT/fuzz_parity.log
SEEDS=30 timeout -k 10 240 python tools/fuzz_api.py > $OUT/fuzz_api.log 2>&1; echo "fuzz_api rc $?"; tail -n 2 $OUT/fuzz_api.log
SEEDS=20 timeout -k 10 200 python tools/fuzz_gsl.py > $OUT/fuzz_gsl.log 2>&1; echo "fuzz_gsl rc $?"; tail -n 2 $OUT/fuzz_gsl.log
SEEDS=30 timeout -k 10 200 python tools/fuzz_last_average.py > $OUT/fuzz_last_average.log 2>&1; echo "fuzz_last_average rc $?"; tail -n 2 $OUT/fuzz_last_average.log
