OUT=gpurun_out/c18
mkdir -p $OUT
SEEDS=400 timeout -k 10 900 python tools/fuzz_canon.py > $OUT/fuzz_canon.log 2>&1; echo "fuzz_canon rc $?"; tail -n 6 $OUT/fuzz_canon.log
python -m pytest tests/test_hip_nshard.py -x -q -m gpu -k "randomised_problems" 2>&1 | tail -n 3
