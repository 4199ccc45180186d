#!/bin/bash
# How many workgroups of a per-patch CAR-HyNet kernel really share a CU?  Kernel durations (rocprofv3) for 256 / 512 / 768 / 1024 patches
# per launch, stagger off: with k co-resident workgroups per CU the duration steps every k x 256 patches.   (through gpurun)
R=$PWD; O=$R/gpurun_out/chocc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 256 512 768 1024 2048; do
  GIMS_CH_STAGGER=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks$n -- python3 $R/tools/carhynet_bench.py --patches $n --reps 3 --no-cpu > $O/log$n 2>&1
  f=$(find $O/ks$n -name '*kernel_stats.csv' | head -1)
  echo "patches $n"
  python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if "ch_conv_block" in r["Name"] or "ch_sandglass" in r["Name"]:
        print("   %-60s avg %8.1f us" % (r["Name"][12:70], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $O/ks$n
done
