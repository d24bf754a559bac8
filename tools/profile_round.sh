#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the default bench under rocprofv3 --kernel-trace --stats, the two PMC
# passes the MI355X guide prescribes for HBM traffic (FETCH_SIZE and WRITE_SIZE cannot share a pass; the DEER
# record is left out of them so that only the two headline sizes launch the strip kernels), then the default
# bench unprofiled with those traffic figures in place.  usage: tools/profile_round.sh <tag>
set -e
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
# the program after `--` must be the interpreter's ELF binary itself (no shim, no `#!/usr/bin/env` hop: the profiler's
# preloaded library has initialised the GPU by then, and an exec after that is forbidden on this pool)
PY=$(readlink -f "$(command -v python3)")
head -c 4 "$PY" | grep -q ELF || { echo "profile_round.sh: $PY is not an ELF binary" >&2; exit 2; }
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $PY bench.py --no-cpu-baseline --no-api --no-one-copy > $OUT/stats.log 2>&1
echo "stats done"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o $TAG -- $PY bench.py --no-cpu-baseline --no-api --no-deer --no-ala5 --no-storage-experiment --no-one-copy --warmup 0 > $OUT/fetch.log 2>&1
echo "fetch done"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o $TAG -- $PY bench.py --no-cpu-baseline --no-api --no-deer --no-ala5 --no-storage-experiment --no-one-copy --warmup 0 > $OUT/write.log 2>&1
echo "write done"
mkdir -p $OUT/profiles
cp profiles/traffic.json $OUT/profiles/ 2>/dev/null || true
PROFILES_OUT=$OUT/profiles python tools/summarize_profiles.py $TAG $OUT/stats $OUT/fetch $OUT/write --key 1000000 1024 --fkey 1000000 512 > $OUT/summary.log
cp $OUT/stats/${TAG}_kernel_stats.csv $OUT/kernel_stats.csv
# the default bench last, with the traffic figures of THESE sources in place (bench.py copies them into roofline.traffic)
cp $OUT/profiles/traffic.json profiles/traffic.json
timeout -k 10 600 python bench.py > $OUT/bench_N1.json 2> $OUT/bench_N1.err
echo "bench done"
# the launch-bound regime: wall time per lock-step round (both engines, one process, interleaved) and the per-kernel split
REPS=4 SIZES=256:100000,1024:125000,205:500000,64:20000,28:50001 VARIANTS=device,host,host-nospec timeout -k 10 300 python tools/engine_ab.py > $OUT/engine_ab.txt 2>&1
# r05: the headline, both engines; and ONE RANK'S SHARE of the headline at 8 / 4 / 2 GPUs (rank 0 of the decomposition alone on
# this GPU, exchanges mirrored): wall per round, then the 8-GPU share's per-kernel timeline
REPS=2 SIZES=1024:1000000 VARIANTS=host,dev1 timeout -k 10 300 python tools/engine_ab.py > $OUT/engine_ab_headline.txt 2>&1
for W in 8 4 2; do WORLD=$W REPS=2 SIZES=1024:1000000 VARIANTS=device timeout -k 10 300 python tools/engine_ab.py >> $OUT/engine_ab_rank_share.txt 2>&1; done
WORLD=8 SIZES=1024:1000000 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/w8 -o w8 -- $PY tools/small_timeline.py > $OUT/w8.log 2>&1
python tools/trace_gaps.py $OUT/w8/w8_kernel_trace.csv > $OUT/round_rank_share_8gpu_1024x1000000.txt
SIZES=1024:1000000 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/big -o big -- $PY tools/small_timeline.py > $OUT/big.log 2>&1
python tools/trace_gaps.py $OUT/big/big_kernel_trace.csv k_trial > $OUT/round_1024x1000000.txt
timeout -k 10 200 python tools/canon_probe.py quick > $OUT/canon_probe.txt 2>&1
SIZES=256:100000 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/small -o small -- $PY tools/small_timeline.py > $OUT/small.log 2>&1
python tools/trace_gaps.py $OUT/small/small_kernel_trace.csv > $OUT/small_round_256x100000.txt
SIZES=1024:125000 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/small2 -o small -- $PY tools/small_timeline.py >> $OUT/small.log 2>&1
python tools/trace_gaps.py $OUT/small2/small_kernel_trace.csv > $OUT/small_round_1024x125000.txt
timeout -k 10 300 python tools/ref_margins.py > $OUT/parity_margins.md 2> $OUT/parity_margins.err
REPS=30 timeout -k 10 200 python tools/strip_probe.py > $OUT/forces_strip_probe.json 2>/dev/null
FORCES_M=1024 REPS=30 timeout -k 10 200 python tools/strip_probe.py > $OUT/forces_strip_probe_M1024.json 2>/dev/null
# matrices taller than 1024 rows: the strip kernels over row panels against the r01 streaming kernels (log-weights sweep)
(SIZES=2048:500000,1536:300000 timeout -k 10 200 python tools/small_timeline.py; echo "-- BIOEN_HIP_PANELS=0 (r01 streaming kernels):"; BIOEN_HIP_PANELS=0 SIZES=2048:500000,1536:300000 timeout -k 10 200 python tools/small_timeline.py) > $OUT/panels.txt 2>&1
echo "timelines done"
rm -f $OUT/*/*_kernel_trace.csv $OUT/*/*_counter_collection.csv     # tens of MB; the summaries are what is kept
tail -1 $OUT/bench_N1.json | cut -c1-400
