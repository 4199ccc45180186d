#!/bin/bash
# Single-pair forward() anatomy on a GPU box (run through gpurun from the repo root):  tools/b1_trace.sh <outdir>
# Per size: rocprofv3 kernel trace of 20 calls -> idle-gap / per-kernel report of one steady-state call; then the latency probe's stage table and
# a cProfile of the host side.
set -u
R=$PWD
O=$R/${1:-gpurun_out/b1}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 1024 4096; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$n -- python3 $R/tools/b1_loop.py $n 20 > $O/loop_$n.log 2>&1
  f=$(find $O/tr_$n -name '*kernel_trace.csv' | head -1)
  python3 $R/tools/gap_report.py $f 6 > $O/gaps_$n.txt 2>&1
  rm -rf $O/tr_$n
done
cd $R
python3 tools/latency_probe.py > $O/latency_probe.txt 2>&1
python3 tools/b1_hostprof.py 1024 > $O/hostprof_1024.txt 2>&1
tail -3 $O/loop_*.log; head -30 $O/gaps_1024.txt; cat $O/latency_probe.txt
