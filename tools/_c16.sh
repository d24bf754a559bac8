set -e
OUT=gpurun_out/c16
mkdir -p $OUT
PY=$(python -c "import sys; print(sys.executable)")
python -m pytest tests/test_hip_nshard.py tests/test_hip_fullsize.py tests/test_hip_parity.py tests/test_hip_multimin.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -n 40 $OUT/tests.log; exit 1; }
tail -n 3 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
WORLD=8 SIZES=1024:1000000 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/w8 -o w8 -- $PY tools/small_timeline.py > $OUT/w8.log 2>&1
python tools/trace_gaps.py $OUT/w8/w8_kernel_trace.csv > $OUT/round_rank_share_8gpu.txt
SIZES=1024:125000 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/small2 -o small -- $PY tools/small_timeline.py > $OUT/small.log 2>&1
python tools/trace_gaps.py $OUT/small2/small_kernel_trace.csv > $OUT/small_round_1024x125000.txt
rm -f $OUT/*/*_kernel_trace.csv
for W in 8; do WORLD=$W REPS=2 SIZES=1024:1000000 VARIANTS=device timeout -k 10 300 python tools/engine_ab.py >> $OUT/engine_ab_rank_share.txt 2>&1; done
cat $OUT/engine_ab_rank_share.txt
head -12 $OUT/round_rank_share_8gpu.txt
SIZES=256:100000 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/small -o small -- $PY tools/small_timeline.py >> $OUT/small.log 2>&1
python tools/trace_gaps.py $OUT/small/small_kernel_trace.csv > $OUT/small_round_256x100000.txt
rm -f $OUT/*/*_kernel_trace.csv
head -12 $OUT/small_round_256x100000.txt
REPS=3 SIZES=256:100000,1024:125000 VARIANTS=device,host timeout -k 10 300 python tools/engine_ab.py
