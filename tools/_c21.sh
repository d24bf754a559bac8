OUT=gpurun_out/c21
mkdir -p $OUT
python -m pytest tests/test_hip_api.py -x -q -m gpu -k "one_strip_copy or cannot_be_allocated" > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -n 15 $OUT/tests.log | cut -c1-300
BIOEN_HIP_ONE_COPY=1 SEEDS=150 timeout -k 10 400 python tools/fuzz_canon.py > $OUT/fuzz_canon_onecopy.log 2>&1; echo "fuzz_canon(one copy) rc $?"; tail -n 3 $OUT/fuzz_canon_onecopy.log | cut -c1-300
BIOEN_HIP_ONE_COPY=1 SEEDS=80 MIN_DIM=4 timeout -k 10 300 python tools/fuzz_parity.py > $OUT/fuzz_parity_onecopy.log 2>&1; echo "fuzz_parity(one copy) rc $?"; tail -n 3 $OUT/fuzz_parity_onecopy.log | cut -c1-300
REPS=2 SIZES=1024:1000000 VARIANTS=host timeout -k 10 300 python tools/engine_ab.py
BIOEN_HIP_ONE_COPY=1 REPS=2 SIZES=1024:1000000 VARIANTS=host timeout -k 10 300 python tools/engine_ab.py
REPS=3 SIZES=256:100000,512:500000 VARIANTS=device timeout -k 10 300 python tools/engine_ab.py
BIOEN_HIP_ONE_COPY=1 REPS=3 SIZES=256:100000,512:500000 VARIANTS=device timeout -k 10 300 python tools/engine_ab.py
