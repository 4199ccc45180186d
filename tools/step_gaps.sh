#!/bin/bash
# Where a batched step leaves the device idle (run through gpurun from the repo root):  tools/step_gaps.sh [kpts pairs [min_gap_us]]
# rocprofv3 kernel trace of a short bench run -> tools/b1_sequence.py on the last full step (the profiler's host overhead widens the gaps it
# shows: read them as WHERE, not how much).
set -u
R=$PWD
K=${1:-4096}; P=${2:-8}; G=${3:-8}
O=$R/gpurun_out/step_gaps
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GIMS_BENCH_NO_STAGE_TIMERS=1      # no event pass: the last step in the trace is a production step
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --kpts $K --pairs $P --steps 6 --warmup 3 --no-cpu-baseline > $O/run.log 2>&1
f=$(find $O/tr -name '*kernel_trace.csv' | head -1)
python3 $R/tools/b1_sequence.py $f $G > $O/gaps_${K}x${P}.txt 2>&1
rm -rf $O/tr
cat $O/gaps_${K}x${P}.txt
