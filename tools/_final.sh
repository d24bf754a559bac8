set -e
OUT=gpurun_out/final
mkdir -p $OUT
python -c "import __graft_entry__ as e; e.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 || { tail -n 20 $OUT/smoke.log; exit 1; }
tail -n 1 $OUT/smoke.log
python -m pytest tests -q -m gpu > $OUT/gpu_tests.txt 2>&1 || { tail -n 40 $OUT/gpu_tests.txt; exit 1; }
tail -n 3 $OUT/gpu_tests.txt
